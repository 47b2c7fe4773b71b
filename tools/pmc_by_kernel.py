"""Aggregate a rocprofv3 counter_collection.csv per kernel: python tools/pmc_by_kernel.py <csv> [name-filter]"""
import csv
import sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-62s n=%-4d %s" % (k, len(cnt[k]), "  ".join("%s=%.4g" % (c, v) for c, v in sorted(d.items()))))
