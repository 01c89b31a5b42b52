"""Time the batched Viterbi through the C ABI on device-resident blocks: python scripts/time_viterbi.py [nblocks] [nsoft]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from jaero_amd import capi

L = capi.lib()
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nsoft = int(sys.argv[2]) if len(sys.argv) > 2 else 4992
g = torch.Generator(device="cuda").manual_seed(1)
soft = torch.randint(0, 256, (nblk, nsoft), device="cuda", dtype=torch.uint8, generator=g)
ov = torch.zeros((nblk, 64), dtype=torch.uint8, device="cuda")
out = torch.zeros((nblk, nsoft // 2), dtype=torch.uint8, device="cuda")
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.check(L.jaero_viterbi_continuous(0, soft.data_ptr(), nblk, nsoft, 24, ov.data_ptr(), out.data_ptr(), None, 1, None))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"call {it}: {dt * 1e3:.3f} ms  ({nblk * nsoft / dt / 1e9:.2f} Gsoftbits/s)")
# the same calls with 2 GiB of unrelated memory traffic in between (what the decoder sees inside the Aero-L pipeline)
big = torch.empty(1 << 31, dtype=torch.uint8, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(4):
    big.add_(1)
    torch.cuda.synchronize()
    ev[0].record()
    capi.check(L.jaero_viterbi_continuous(0, soft.data_ptr(), nblk, nsoft, 24, ov.data_ptr(), out.data_ptr(), None, 1, None))
    ev[1].record()
    torch.cuda.synchronize()
    print(f"after 2 GiB of other traffic, call {it}: {ev[0].elapsed_time(ev[1]):.3f} ms")
for it in range(3):
    torch.cuda.synchronize()
    ev[0].record()
    capi.check(L.jaero_viterbi_continuous(0, soft.data_ptr(), nblk, nsoft, 24, ov.data_ptr(), out.data_ptr(), None, 1, None))
    ev[1].record()
    torch.cuda.synchronize()
    print(f"back to back, call {it}: {ev[0].elapsed_time(ev[1]):.3f} ms")
