#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch.
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [substring-filter]"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt and flt not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            print("   %-34s %16.1f  (x%d)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))


if __name__ == "__main__":
    main()
