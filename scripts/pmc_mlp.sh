#!/bin/bash
# PMC passes for the MLP kernel (run on the GPU box via gpurun).  Counters in separate runs, no tracing domains.
# usage: [PREC=bf16x3] [KERNEL=regex] [SCRIPT=scripts/time_train_step.py] bash scripts/pmc_mlp.sh <outdir-name> [quick|sq]
# (SCRIPT: profile that script instead of bench.py, e.g. the training-style step for the backward GEMMs)
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters
  if [ -n "$SCRIPT" ]; then
    rocprofv3 --pmc $2 --kernel-include-regex "${KERNEL:-mlp_fwd}" --output-format csv -d $OUT/$1 -- python3 $R/$SCRIPT > $OUT/$1.log 2>&1
  else
    rocprofv3 --pmc $2 --kernel-include-regex "${KERNEL:-mlp_fwd}" --output-format csv -d $OUT/$1 -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-rays 0 --no-reuse --precision ${PREC:-f32} > $OUT/$1.log 2>&1
  fi
  echo "$1 rc=$?"
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
if [ "$2" != "quick" ]; then
run sq2 "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
if [ "$2" != "sq" ]; then
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
fi
fi
