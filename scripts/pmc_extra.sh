#!/bin/bash
# Extra SQ counters for the bf16x3 MLP kernel (MFMA / VALU co-execution, LDS conflicts): bash scripts/pmc_extra.sh <outdir-name>
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_extra}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  rocprofv3 --pmc $2 --kernel-include-regex "${KERNEL:-mlp_fwd_bf16}" --output-format csv -d $OUT/$1 -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-rays 0 --no-reuse --precision ${PREC:-bf16x3} > $OUT/$1.log 2>&1
  echo "$1 rc=$?"
}
run coexec "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_LOAD"
