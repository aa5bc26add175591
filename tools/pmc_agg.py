"""Mean of every counter per kernel from rocprofv3 --pmc output: python3 tools/pmc_agg.py <dir> [name-substring]"""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if sub in k:
        print(k[:110])
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"    {c:26s} n={len(v):3d} mean={sum(v) / len(v):.5g}")
