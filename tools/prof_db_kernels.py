"""Per-kernel statistics from a rocprofv3 (rocpd sqlite) database: calls, mean / median / min duration in us.
usage: prof_db_kernels.py results.db [name_substring]"""
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, start, end from kernels").fetchall() if False else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_c = "name" if "name" in cols else cols[0]
q = f"select {name_c}, start, end from kernels"
by = {}
for name, s, e in db.execute(q):
    if want in name:
        by.setdefault(name, []).append((e - s) / 1e3)
tot = sum(sum(v) for v in by.values())
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:100]:100s} n={len(v):5d} mean={statistics.mean(v):9.2f} med={statistics.median(v):9.2f} "
          f"min={min(v):9.2f} total_ms={sum(v) / 1e3:8.2f} ({100 * sum(v) / tot:4.1f}%)")
