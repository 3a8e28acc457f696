"""Compact a rocprofv3 --kernel-trace --stats kernel_stats.csv (kernel names truncated)."""

import csv
import sys


def main(src, dst):
  rows = list(csv.DictReader(open(src)))
  with open(dst, "w") as f:
    f.write("kernel,calls,total_ms,avg_us,pct\n")
    for r in rows:
      name = r["Name"].split("(")[0].replace("void ", "")[:70]
      f.write(f'"{name}",{r["Calls"]},{float(r["TotalDurationNs"]) / 1e6:.3f},{float(r["AverageNs"]) / 1e3:.2f},{float(r["Percentage"]):.2f}\n')


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2])
