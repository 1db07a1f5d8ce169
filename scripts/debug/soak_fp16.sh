cd $GRAFT_REPO_ROOT
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 1100 python "$@" > gpurun_out/soakh_$name.log 2>&1; echo "   exit $?: $(grep -E '^fuzz_' gpurun_out/soakh_$name.log | tail -1 | cut -c1-500)"; }
run dropin tests/fuzz_dropin.py --cases 400 --steps 60 --seed 31
export UCNERF_SPLIT_OPERAND=fp16
run render tests/fuzz_render.py --cases 700 --seed 11 --out gpurun_out/soakh_render.json
run pipeline tests/fuzz_pipeline.py --cases 800 --steps 40 --seed 31
run mlp tests/fuzz_mlp.py --cases 400 --seed 11
