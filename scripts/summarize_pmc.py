#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/pmc_summary*.json.

usage: summarize_pmc.py <dir with pmc_FETCH_SIZE.csv and pmc_WRITE_SIZE.csv> <tag> <channels_per_gpu> [out.json]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, averaged over the FULL launches of a kernel (a step's sample-loop work
is one full launch plus a one-sample launch behind the coarse estimate; the latter are dropped: values < 10 % of the largest).
The factors come from scripts/ubench/hbm_counters.hip run under the same two passes on the same GPU (profiles/r2_counter_calibration.json):
FETCH_SIZE reports exactly half of the bytes of 8-byte-per-lane and of 16-byte-per-lane row reads, WRITE_SIZE the bytes of row
writes exactly; a store of 16 bytes into a lane's own 64-byte sector that is completed by the next three stores costs 1.19 x."""
import collections
import csv
import json
import sys

d, tag, nch = sys.argv[1], sys.argv[2], int(sys.argv[3])
out = sys.argv[4] if len(sys.argv) > 4 else None
vals = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    r = collections.defaultdict(list)
    for row in csv.DictReader(open(f"{d}/pmc_{c}.csv")):
        r[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in r.items():
        m = max(v)
        w = [x for x in v if x >= 0.1 * m]
        vals[k][c + "_KiB_mean"] = sum(w) / len(w)
        vals[k]["full_launches_" + c] = len(w)
summ = {"tag": tag, "channels_per_gpu": nch,
        "correction": "gfx950: (2*FETCH_SIZE + WRITE_SIZE)*1024, factors calibrated with scripts/ubench/hbm_counters.hip (profiles/r2_counter_calibration.json)"}
for k, v in vals.items():
    if "FETCH_SIZE_KiB_mean" in v and "WRITE_SIZE_KiB_mean" in v:
        v["kernel"] = k
        v["hbm_read_bytes_per_launch"] = 2 * v["FETCH_SIZE_KiB_mean"] * 1024
        v["hbm_write_bytes_per_launch"] = v["WRITE_SIZE_KiB_mean"] * 1024
        v["hbm_bytes_per_launch"] = v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]
        summ[k] = v
        if "k_oqpsk" in k or "k_msk" in k:
            summ["sample_loop"] = v
        if "k_coarse" in k:
            summ["coarse_freq"] = v
txt = json.dumps(summ, indent=1)
if out:
    open(out, "w").write(txt + "\n")
print(txt)
