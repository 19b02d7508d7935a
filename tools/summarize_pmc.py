#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch.
usage: tools/summarize_pmc.py <profile dir> [kernel substring]"""
import csv, glob, os, sys, collections

def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "extend_vote_kernel"
    for path in sorted(glob.glob(os.path.join(root, "pmc_*", "*", "*_counter_collection.csv"))):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))   # counter -> dispatch -> sum
        for row in csv.DictReader(open(path)):
            if filt not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        print(f"## {os.path.relpath(path, root)}  (kernel ~ '{filt}')")
        for name, d in sorted(acc.items()):
            vals = sorted(d.values())
            # the set-up launch (primary rays only) is the smallest dispatch: report the bench launches (largest)
            top = vals[-3:] if len(vals) > 3 else vals
            print(f"{name:32s} dispatches={len(vals):2d}  mean(bench launches)={sum(top)/len(top):.6g}  all={['%.4g' % v for v in vals]}")

if __name__ == "__main__":
    main()
