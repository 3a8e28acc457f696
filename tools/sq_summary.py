"""Summarise tools/sq_counters.sh output: mean counter value per kernel launch."""
import collections
import csv
import glob
import sys

tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}/p*/pmc_counter_collection.csv"):
  for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_"):
      acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
  print(k)
  for c, v in sorted(d.items()):
    print(f"   {c:34s} {sum(v) / len(v):16.0f}   (n={len(v)})")
