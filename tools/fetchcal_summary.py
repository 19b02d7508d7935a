#!/usr/bin/env python3
"""Summarise a `rocprofv3 --pmc FETCH_SIZE` run of tools/bin/ubench_fetchcal: counter bytes against known bytes.
usage: tools/fetchcal_summary.py <rocprof output dir> <ubench stdout file>"""
import collections, csv, glob, os, sys

def main():
    root, log = sys.argv[1], sys.argv[2]
    known = {}
    for line in open(log):
        w = line.split()
        if len(w) == 3 and w[0] == "known_bytes":
            known[w[1]] = int(w[2])
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(dict)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0]
            acc[(name, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
            if "Start_Timestamp" in row and row.get("End_Timestamp"):
                dur[name][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    print(f"{'kernel':18s} {'counter':12s} {'known MiB':>10s} {'counter KiB -> MiB':>20s} {'known / counter':>16s} {'ms (profiled)':>14s}")
    for (name, ctr), d in sorted(acc.items()):
        if name not in known:
            continue
        v = sorted(d.values())[-1]        # the passes repeat; take one
        ms = sorted(dur[name].values())[0] if dur.get(name) else float("nan")
        print(f"{name:18s} {ctr:12s} {known[name] / 2**20:10.1f} {v / 1024:20.1f} {known[name] / (v * 1024):16.3f} {ms:14.3f}")

if __name__ == "__main__":
    main()
