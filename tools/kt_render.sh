#!/bin/bash
# kernel trace of the render leg only.  usage: tools/kt_render.sh <outdir> [bench_render args]
OUT=${1:-gpurun_out/ktr}; shift || true
ARGS="${*:---scene c3 --spp 32}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 tools/bench_render.py $ARGS > "$OUT/kt.log" 2>&1 || { echo "kt failed"; tail -5 "$OUT/kt.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-70s calls %5s  total %9.2f ms  avg %8.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
