"""Sums rocprofv3 --pmc counter rows per kernel name substring: python tools/pmc_summary.py <counter_collection.csv> <substr>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in rows:
    if sub in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print("%-28s calls %3d  total %.4g  per call %.4g" % (k, n[k], acc[k], acc[k] / n[k]))
