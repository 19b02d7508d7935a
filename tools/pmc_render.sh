#!/bin/bash
# Kernel trace + memory counters of the render leg (tools/bench_render.py).  usage: tools/pmc_render.sh <outdir> [bench_render args]
set -u
OUT=${1:-gpurun_out/pmcr}; shift || true
ARGS="${*:---scene c3 --spp 32}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 tools/bench_render.py $ARGS > "$OUT/kt.log" 2>&1 || { echo "kt failed"; tail -5 "$OUT/kt.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/bench_render.py $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { echo "pmc_fetch failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d "$OUT/pmc_write" -- python3 tools/bench_render.py $ARGS > "$OUT/pmc_write.log" 2>&1 || { echo "pmc_write failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
python3 tools/summarize_pmc_all.py "$OUT"
