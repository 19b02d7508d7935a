#!/bin/bash
# SQ / TA / TD utilisation counters of the render leg.  usage: tools/pmc_render_sq.sh <outdir> [bench_render args]
set -u
OUT=${1:-gpurun_out/pmcrsq}; shift || true
ARGS="${*:---scene c3 --spp 16}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 tools/bench_render.py $ARGS > "$OUT/kt.log" 2>&1 || { echo "kt failed"; tail -5 "$OUT/kt.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_sq1" -- python3 tools/bench_render.py $ARGS > "$OUT/pmc_sq1.log" 2>&1 || { echo "pmc_sq1 failed"; tail -5 "$OUT/pmc_sq1.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/pmc_mem" -- python3 tools/bench_render.py $ARGS > "$OUT/pmc_mem.log" 2>&1 || { echo "pmc_mem failed"; tail -5 "$OUT/pmc_mem.log"; exit 1; }
python3 tools/summarize_pmc_all.py "$OUT"
