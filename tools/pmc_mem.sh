#!/bin/bash
# Two counter passes (issue mix + vector-memory path) over the closest-hit bench with the current PTMI_* environment.
# usage: PTMI_EXTEND_VARIANT=3 tools/pmc_mem.sh <outdir> <kernel substring>
set -u
OUT=${1:-gpurun_out/pmcm}; KERN=${2:-extend_}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render --no-traffic --no-hbm-leg > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || { echo "pmc $name failed"; tail -5 "$OUT/pmc_$name.err"; exit 1; }
}
run issue SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
run mem TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY
python3 tools/summarize_pmc.py "$OUT" "$KERN"
