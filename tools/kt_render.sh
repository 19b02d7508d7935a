#!/bin/bash
# Per-kernel times of a render (rocprofv3 --kernel-trace --stats).  usage: tools/kt_render.sh <outdir> [bench_render args]
set -u
OUT=${1:-gpurun_out/kt}; shift || true
ARGS="${*:---scene c3 --spp 256}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 tools/bench_render.py $ARGS > "$OUT/kt.log" 2>&1 || { echo "kt failed"; tail -5 "$OUT/kt.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for p in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(out + "/kernel_stats.txt", "w") as f:
    for r in rows[:24]:
        line = f'{r["Name"][:90]:90s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e6:9.2f} ms avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):5.1f}%'
        print(line); f.write(line + "\n")
PY
tail -2 "$OUT/kt.log"
