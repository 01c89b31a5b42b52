#!/usr/bin/env python3
"""Copy what is to be judged from gpurun_out/<tag>/ (scratch) into profiles/ (tracked): bench lines, kernel stats, counter summaries, test
logs -- as profiles/<tag>_<name>; the counter summaries bench.py reads (pmc_summary*.json, sq_summary*.json) also under their plain names.
usage: scripts/collect_evidence.py <tag>"""
import glob
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
keep = ["bench_line*.json", "bench_wall.txt", "kernel_stats*.csv", "pmc_summary*.json", "sq_summary*.json", "pytest_gpu.log", "smoke.log"]
n = 0
for pat in keep:
    for f in sorted(glob.glob(os.path.join(src, pat))):
        if os.path.getsize(f) == 0:
            continue
        b = os.path.basename(f)
        shutil.copy(f, os.path.join(dst, f"{tag}_{b}"))
        if b.startswith(("pmc_summary", "sq_summary")):
            shutil.copy(f, os.path.join(dst, b))
        n += 1
print(f"{n} files from gpurun_out/{tag}/ -> profiles/")
