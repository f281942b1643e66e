"""Timeline of the last update in a rocprofv3 --kernel-trace CSV of tools/trace_train.py: every dispatch from the last
k_rollout on, with its start relative to that launch, duration and the gap to the previous kernel's end (us)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_rollout" in r["Kernel_Name"])
prev = max(i for i, r in enumerate(rows[:last]) if "k_rollout" in r["Kernel_Name"])
t0, end = int(rows[prev]["Start_Timestamp"]), None
for r in rows[prev:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end) / 1e3 if end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.2f}  gap {gap:7.2f}  {r['Kernel_Name'][:110]}")
    end = e
