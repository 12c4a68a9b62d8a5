#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name: mean counter value per dispatch."""
import csv, glob, sys, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"(k_[a-z0-9_]+)", name)                          # mangled or demangled, with or without (anonymous namespace)
        short = m.group(1) if m else name[:30]
        if "IfL" in name or "If" in name.split("E")[0][-3:]: short += "<f32>"
        rows[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in rows for c in rows[k]})
print("kernel".ljust(22) + "n".rjust(5) + "".join(c.replace("SQ_", "").rjust(22) for c in names))
for k in sorted(rows):
    n = max(len(v) for v in rows[k].values())
    print(k.ljust(22) + str(n).rjust(5) + "".join((f"{sum(rows[k][c]) / len(rows[k][c]):.3e}" if rows[k][c] else "-").rjust(22) for c in names))
