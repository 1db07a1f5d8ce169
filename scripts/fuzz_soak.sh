#!/bin/bash
# Long runs of the fuzzers under tests/ with seeds the fixed sets and the earlier runs did not use: scripts/fuzz_soak.sh <part 1|2|3>
# (three parts so that each fits one gpurun call; logs and one summary line per fuzzer under gpurun_out/soak_*.log)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 1100 python "$@" > gpurun_out/soak_$name.log 2>&1; echo "   exit $?: $(grep -E '^fuzz_' gpurun_out/soak_$name.log | tail -1 | cut -c1-400)"; }
case "$1" in
  1) run dropin tests/fuzz_dropin.py --cases 2000 --steps 60 --seed 11
     run pipeline tests/fuzz_pipeline.py --cases 3000 --steps 40 --seed 11
     run sampling tests/fuzz_sampling.py --cases 5000 --seed 11
     run builders tests/fuzz_builders.py --cases 1200 --seed 11 ;;
  2) run render tests/fuzz_render.py --cases 1100 --seed 11 --out gpurun_out/soak_render.json
     run mlp tests/fuzz_mlp.py --cases 1500 --seed 11 ;;
  3) run grads tests/fuzz_grads.py --cases 330 --seed 11
     run mvs tests/fuzz_mvs.py --cases 350 --seed 11 ;;
  *) echo "usage: $0 <1|2|3>"; exit 2 ;;
esac
