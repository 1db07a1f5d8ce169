#!/bin/bash
# Counters of the cost-volume kernels at the three cascade-stage shapes: bash scripts/pmc_mvs.sh [outdir-name]
# Separate rocprofv3 --pmc passes (no tracing domains), summarised per kernel AND per launch shape by scripts/pmc_mvs_summary.py.
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_mvs}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  timeout -k 10 150 rocprofv3 --pmc $2 --kernel-include-regex "cost_volume" --output-format csv -d $OUT/$1 -- python3 $R/scripts/time_mvs_stage.py > $OUT/$1.log 2>&1
  echo "$1 rc=$?"
}
run sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
run sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
run ta1 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
run tcp1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
run tcp2 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
python3 $R/scripts/pmc_mvs_summary.py $OUT $OUT/summary.json
