#!/bin/bash
# The round's closing measurements on one box: bench line, per-kernel-class times of the named scenes (one lane, the
# library's own events), rocprofv3 kernel trace of C3 (one lane, so the kernels do not overlap), SQ passes of C3 and C4.
# usage: tools/final_profiles.sh <outdir>
set -u
OUT=${1:-gpurun_out/final}; mkdir -p "$OUT"
timeout -k 10 500 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
{
  for args in "--scene c3 --spp 256" "--scene c3 --spp 64 --bounces 4" "--scene c4 --width 3840 --height 2160 --spp 32" \
              "--scene c5 --width 3840 --height 2160 --spp 2" "--scene c1 --width 512 --height 512 --spp 16"; do
    echo "== $args"
    timeout -k 10 300 python3 tools/bench_render.py $args 2>&1 | grep -E "Msamples" || exit 1
    timeout -k 10 300 python3 tools/bench_render.py $args --profile 2>&1 | grep -E "profile" || exit 1
  done
} > "$OUT/render_scenes.txt" 2>&1 || { echo "render scenes failed"; tail -5 "$OUT/render_scenes.txt"; exit 1; }
export PTMI_RENDER_LANES=1
bash tools/kt_render.sh "$OUT/kt_c3" --scene c3 --spp 256 > "$OUT/kt_c3.txt" 2>&1 || { echo "kt failed"; exit 1; }
unset PTMI_RENDER_LANES
bash tools/pmc_render_sq.sh "$OUT/sq_c3" --scene c3 --spp 64 > "$OUT/render_c3_sq.txt" 2>&1 || { echo "sq c3 failed"; exit 1; }
bash tools/pmc_render_sq.sh "$OUT/sq_c4" --scene c4 --width 3840 --height 2160 --spp 8 > "$OUT/render_c4_sq.txt" 2>&1 || { echo "sq c4 failed"; exit 1; }
echo done
