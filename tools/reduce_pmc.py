"""Shrink a rocprofv3 pmc_counter_collection.csv in place to one row per (kernel, counter) holding the mean
over launches (the per-dispatch file of a multi-counter pass exceeds what gpurun copies back)."""
import collections
import csv
import sys

path = sys.argv[1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
  acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(path, "w", newline="") as f:
  w = csv.writer(f)
  w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Launches"])
  for (k, c), v in sorted(acc.items()):
    w.writerow([k, c, sum(v) / len(v), len(v)])
