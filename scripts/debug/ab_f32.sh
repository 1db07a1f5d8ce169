R=$GRAFT_REPO_ROOT
for n in 512 1536; do for v in f32old -; do
 if [ "$v" = "-" ]; then unset UCNERF_LIB; else export UCNERF_LIB=$R/uc_nerf_amd/libucnerf_hip_$v.so; fi
 timeout -k 10 200 python $R/bench.py --precision f32 --rays $n --no-reuse --cpu-rays 0 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('f32 rays $n lib $v ms/step %.4f' % d['ms_per_step'])"
done; done
