cd $GRAFT_REPO_ROOT
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 1100 python "$@" > gpurun_out/soak4_$name.log 2>&1; echo "   exit $?: $(grep -E '^fuzz_' gpurun_out/soak4_$name.log | tail -1 | cut -c1-400)"; }
run dropin tests/fuzz_dropin.py --cases 1500 --steps 80 --seed 12
run pipeline tests/fuzz_pipeline.py --cases 3000 --steps 40 --seed 12
run builders tests/fuzz_builders.py --cases 1500 --seed 12
run mlp tests/fuzz_mlp.py --cases 1000 --seed 12
