#!/usr/bin/env python3
"""Summarise one gpurun_out/<tag>/ directory written by scripts/gpu_round.sh.

* kernel stats: our kernels' rows of rocprofv3's *_kernel_stats.csv (names truncated to 100 chars) -> kernel_stats.csv
* PMC: per-kernel mean FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them) and the HBM bytes per launch derived
  the way MI355X_MICROARCH.md "HBM" prescribes for gfx950: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE
  tallies 128-B read requests at 64 B on this rocprofv3; WRITE_SIZE is taken as reported) -> pmc_summary.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
out = sys.argv[1]
OURS = ("k_oqpsk", "k_msk", "k_coarse", "k_viterbi", "k_transpose", "k_center", "k_status", "k_pack", "k_burst", "k_aerol", "k_hilbert", "k_trident", "k_hist")


def short(name):
    name = name.split("(")[0]
    return name[:100]


def role(name):
    if "k_oqpsk_samples" in name or "k_msk_samples" in name:
        return "sample_loop"
    if "k_coarse" in name:
        return "coarse_freq"
    for k, r in (("k_burst_front", "front"), ("k_hilbert", "hilbert"), ("k_trident", "trident"), ("k_burst_oqpsk_demod", "demod"), ("k_burst_msk_fb", "demod")):
        if k in name:
            return r
    return None


rows = []
for f in glob.glob(os.path.join(out, "prof", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(r)
if rows:
    keys = list(rows[0].keys())
    with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as fo:
        w = csv.DictWriter(fo, keys)
        w.writeheader()
        for r in rows:
            r = dict(r)
            r["Name"] = short(r["Name"])
            w.writerow(r)
    print("kernel stats (ours):")
    for r in rows:
        if any(k in r["Name"] for k in OURS):
            print("  %-70s calls %6s avg %12.1f us  %5s%%" % (short(r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))

pmc = {}
for which in ("fetch", "write", "rdreq"):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc_" + which, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if any(k in n for k in OURS):
                acc[(short(n), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (n, c), v in acc.items():
        # drop warm-up launches of the sample loop that ran on a partially filled ring: use the upper half by value
        v = sorted(v)
        v = v[len(v) // 2:]
        pmc.setdefault(n, {})[c] = sum(v) / len(v)
        pmc[n]["launches_" + which] = len(acc[(n, c)])
summary = {}
for n, d in pmc.items():
    f, w = d.get("FETCH_SIZE"), d.get("WRITE_SIZE")
    rd, rd32 = d.get("TCC_EA0_RDREQ_sum"), d.get("TCC_EA0_RDREQ_32B_sum")
    e = {"kernel": n, "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w, "TCC_EA0_RDREQ_sum_mean": rd, "TCC_EA0_RDREQ_32B_sum_mean": rd32}
    if f is None and rd is not None:
        # FETCH_SIZE's own gfx950 expression without the TCC_BUBBLE term (counter_defs.yaml): 64 B per request, 32 B for the 32B ones
        f = ((rd - (rd32 or 0.0)) * 64.0 + (rd32 or 0.0) * 32.0) / 1024.0
        e["FETCH_SIZE_KiB_from_RDREQ"] = f
    if f is not None and w is not None:
        e["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        e["correction"] = "gfx950: (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE tallies 128-B read requests at 64 B; MI355X_MICROARCH.md HBM)"
    summary[n] = e
    r = role(n)
    if r and "hbm_bytes_per_launch" in e and (r not in summary or e["hbm_bytes_per_launch"] > summary[r].get("hbm_bytes_per_launch", 0)):
        summary[r] = dict(e)
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
