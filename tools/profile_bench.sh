#!/bin/bash
# Profile bench.py on the GPU box: one kernel-trace pass + separate PMC passes (rocprofv3 does not
# split counters over passes; each block's slot limit is respected per pass -- MI355X_MICROARCH.md).
# usage: tools/profile_bench.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/prof}; shift || true
EXTRA_ARGS="$*"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-render --no-traffic $*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() {  # name, rocprof options...
  local name=$1; shift
  local args="$ARGS"
  # the kernel-trace pass runs the bench's default steps / warm-up, so its kernel averages are those of the bench line
  if [ "${KT:-0}" = 1 ]; then args="--no-cpu-baseline --no-render --no-traffic $EXTRA_ARGS"; fi
  timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $args \
      > "$OUT/$name.json" 2> "$OUT/$name.err"
  local rc=$?
  echo "$name rc=$rc"
  return $rc
}
KT=1 run kt --kernel-trace --stats || exit 1
run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LEVEL_WAVES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
run pmc_mem --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
python3 tools/summarize_pmc.py "$OUT" "extend_vote_kernel<24, false>" > "$OUT/pmc_summary.txt" 2>&1
python3 tools/summarize_pmc.py "$OUT" "extend_vote_kernel<28, false>" > "$OUT/pmc_summary_hbm_leg.txt" 2>&1
cp "$OUT"/kt/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
echo done
