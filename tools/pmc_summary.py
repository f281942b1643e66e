"""Mean per-dispatch value of every counter in a rocprofv3 --pmc counter_collection.csv, for
kernels whose name contains argv[2]."""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:34s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
