"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count, mean/median/min duration (us)."""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in by.values())
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{k[:100]:100s} n={len(v):5d} mean={statistics.mean(v):9.2f} med={statistics.median(v):9.2f} min={min(v):9.2f} total_ms={sum(v)/1e3:8.2f} ({100*sum(v)/tot:4.1f}%)")
