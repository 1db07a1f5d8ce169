#!/bin/bash
# Vector-memory path counters (TA / TCP / TD) for the channel-last gather: bash scripts/pmc_gather.sh <outdir-name>
# At most two counters of a block per run (more: "exceeds the capabilities of the hardware"), no tracing domains.  Summarise with scripts/pmc_by_kernel.py or read the counter_collection csv.
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_gather}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  timeout -k 10 150 rocprofv3 --pmc $2 --kernel-include-regex "${KERNEL:-feat_gather_cl}" --output-format csv -d $OUT/$1 -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-rays 0 --no-reuse > $OUT/$1.log 2>&1
  echo "$1 rc=$?"
}
run ta1 "TA_TA_BUSY_sum GRBM_GUI_ACTIVE"
run ta2 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
run ta3 "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum"
run tcp1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
run tcp2 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
run tcp3 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
run tcp4 "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
run td "TD_TD_BUSY_sum"
run sq "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"
