#!/bin/bash
# One SQ counter pass over the closest-hit bench (no render leg) with the current PTMI_* environment.
# usage: PTMI_EXTEND_VARIANT=3 tools/pmc_quick.sh <outdir> <kernel substring>
set -u
OUT=${1:-gpurun_out/pmcq}; KERN=${2:-extend_}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU \
    --output-format csv -d "$OUT/pmc_sq1" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-render > "$OUT/pmc_sq1.json" 2> "$OUT/pmc_sq1.err" || { echo "pmc failed"; tail -5 "$OUT/pmc_sq1.err"; exit 1; }
python3 tools/summarize_pmc.py "$OUT" "$KERN"
