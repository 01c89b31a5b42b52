#!/usr/bin/env python3
"""Instruction mix of a line range of an AMDGPU .s listing (hipcc -save-temps): how many wave instructions of each class, and by
basic block.  usage: isa_mix.py file.s first_line last_line [--blocks]"""
import collections
import re
import sys

f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
blocks = "--blocks" in sys.argv
lines = open(f).read().split("\n")[a - 1:b]


def cls(m):
    if m.startswith("v_accvgpr") or m in ("v_mov_b32", "v_mov_b64"):
        return "vmov/acc"
    if m.startswith("v_") and m.endswith("_f64") or "_f64_" in m or m in ("v_fmac_f64", "v_fma_f64"):
        if m.startswith(("v_rcp", "v_rsq", "v_sqrt")):
            return "valu_f64_trans"
        return "valu_f64"
    if m.startswith("v_"):
        return "valu_other"
    if m.startswith("s_waitcnt"):
        return "waitcnt"
    if m.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if m.startswith("s_nop"):
        return "nop"
    if m.startswith("s_load") or m.startswith("s_buffer"):
        return "smem"
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("scratch_"):
        return "scratch"
    if m.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


tot = collections.Counter()
per = collections.OrderedDict()
cur = "(entry)"
mn = collections.Counter()
for ln in lines:
    s = ln.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        cur = m.group(1)
        continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        continue
    op = s.split()[0]
    c = cls(op)
    tot[c] += 1
    mn[op] += 1
    per.setdefault(cur, collections.Counter())[c] += 1
n = sum(tot.values())
print("total", n, dict(tot.most_common()))
if blocks:
    for k, v in per.items():
        print(f"{k:12s} {sum(v.values()):5d}", dict(v.most_common()))
print("top mnemonics:", mn.most_common(45))
