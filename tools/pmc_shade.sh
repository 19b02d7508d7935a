#!/bin/bash
# Where the integrator kernels' wave-cycles go: wait / issue breakdown + memory counters.  usage: tools/pmc_shade.sh <outdir> [bench_render args]
set -u
OUT=${1:-gpurun_out/pmcshade}; shift || true
ARGS="${*:---scene c3 --spp 64}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 tools/bench_render.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "$name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
run kt --kernel-trace --stats
run pmc_wait --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run pmc_inst --pmc FETCH_SIZE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES SQ_BUSY_CYCLES
python3 tools/summarize_pmc_all.py "$OUT"
