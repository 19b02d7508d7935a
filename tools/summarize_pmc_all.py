#!/usr/bin/env python3
"""Per kernel: calls, total ms (kernel trace) and per-launch means of every PMC counter found under <dir>/pmc_*.
usage: tools/summarize_pmc_all.py <profile dir>"""
import csv, glob, os, sys, collections, re

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.split(r"\(", name)[0][:48]

def main():
    root = sys.argv[1]
    stats = {}
    for path in glob.glob(os.path.join(root, "kt", "*", "*kernel_stats.csv")):
        for r in csv.DictReader(open(path)):
            stats[short(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    counters = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for path in sorted(glob.glob(os.path.join(root, "pmc_*", "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            counters[short(r["Kernel_Name"])][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k in sorted(stats, key=lambda k: -stats[k][1]):
        calls, ms = stats[k]
        print(f"## {k}: {calls} launches, {ms:.2f} ms total, {ms / calls * 1e3:.1f} us avg")
        for c, d in sorted(counters.get(k, {}).items()):
            vals = list(d.values())
            print(f"   {c:32s} mean/launch {sum(vals) / len(vals):.5g}   total {sum(vals):.5g}")

if __name__ == "__main__":
    main()
