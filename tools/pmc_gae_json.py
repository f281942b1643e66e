"""Summarise the three rocprofv3 --pmc passes of tools/pmc_gae_packed.sh into the JSON bench.py reads
(profiles/r03_gae_packed_pmc_traffic.json)."""
import csv
import glob
import json
import sys

root = sys.argv[1]
T, N = 128, 65536


def mean_counter(sub, counter, kernel_part):
    vals = []
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_part in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    assert vals, (sub, counter, kernel_part)
    return sum(vals) / len(vals), len(vals)


def n_trunc(sub):
    for line in open(f"{root}/{sub}.log"):
        if line.startswith("n_trunc per launch:"):
            return int(line.split(":")[1])
    raise SystemExit("no n_trunc line")


fetch_kb, n = mean_counter("fetch", "FETCH_SIZE", "k_gae_words")
write_kb, _ = mean_counter("write", "WRITE_SIZE", "k_gae_words")
calib_kb, _ = mean_counter("calib", "FETCH_SIZE", "k_gae")
nt = n_trunc("fetch")
known = 20 * T * N                       # the dense i32 form reads r, v, ev, done, trunc in full
fetch = 2 * fetch_kb * 1024
algorithmic = int(12.25 * T * N + 4 * nt)
print(json.dumps({
    "T": T, "N": N, "kernel": "k_gae_words<1,3,true> (ppo_gae_dense_packed)", "launches": n,
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "fetch_bytes_corrected": fetch, "write_bytes": write_kb * 1024,
    "hbm_bytes_per_launch": fetch + write_kb * 1024,
    "algorithmic_bytes_per_launch": algorithmic,
    "traffic_over_algorithmic": (fetch + write_kb * 1024) / algorithmic,
    "truncated_cells_per_launch": nt,
    "calibration": {"dense k_gae<1,8,true> FETCH_SIZE_KB": calib_kb, "known_read_bytes": known,
                    "x2_corrected_bytes": 2 * calib_kb * 1024, "ratio": 2 * calib_kb * 1024 / known},
    "method": "separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over tools/pmc_gae.py --packed (12 launches, "
              "4 rotating buffer sets); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B), "
              "the doubling checked in the same run on the dense i32 form, whose read bytes are known and whose loads have "
              "the same width (4 B per lane); tools/pmc_gae_packed.sh",
}, indent=1))
