#!/bin/bash
# HBM-bound leg: tools/bench_incoherent.py plain, then under the two PMC passes the guide prescribes.
# usage: tools/pmc_incoherent.sh <outdir>
set -u
OUT=${1:-gpurun_out/incoherent}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 400 python3 tools/bench_incoherent.py > "$OUT/plain.txt" 2>&1 || { echo "plain run failed"; tail -5 "$OUT/plain.txt"; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/bench_incoherent.py --steps 3 > "$OUT/pmc_fetch.log" 2>&1 || { echo "pmc_fetch failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 tools/bench_incoherent.py --steps 3 > "$OUT/pmc_write.log" 2>&1 || { echo "pmc_write failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
vals = {}
for p in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    d = {}
    for r in csv.DictReader(open(p)):
        if "extend_vote_kernel" not in r["Kernel_Name"]:
            continue
        d.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
        d[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for name, disp in d.items():
        ids = sorted(disp)[-3:]                       # the timed launches (the counted and the warm-up launch come first)
        vals[name] = sum(disp[i] for i in ids) / len(ids)
plain = open(out + "/plain.txt").read()
ms = float(re.search(r"([\d.]+) ms per launch", plain).group(1))
alg = float(re.search(r"algorithmic ([\d.]+) GB per launch", plain).group(1))
f, w = vals["FETCH_SIZE"] * 1024 / 1e9, vals["WRITE_SIZE"] * 1024 / 1e9
hit = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
lines = [plain.strip(),
         f"PMC (mean of 3 launches): FETCH_SIZE {f:.2f} GB, WRITE_SIZE {w:.2f} GB per launch; L2 hit rate {hit:.3f}",
         f"fabric traffic per launch: {f + w:.2f} GB as counted (exact for 64-byte gathers, the dominant pattern here) .. {2 * f + w:.2f} GB with FETCH_SIZE doubled "
         f"(the guide's correction for 128-byte requests)",
         f"hbm_frac_counter = {(f + w) / ms / 8.0:.3f} .. {(2 * f + w) / ms / 8.0:.3f} of the 8 TB/s peak ({(f + w) / ms:.2f} .. {(2 * f + w) / ms:.2f} TB/s); "
         f"traffic / algorithmic = {(f + w) / alg:.2f} .. {(2 * f + w) / alg:.2f}"]
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
