"""First measurement script for the Aero-L C-channel bit pipeline (AeroL::DecodeC, row f4): frames per second of a bank fed one 4200-bit
frame per channel and step from host memory staged once (the P-channel bench's shape).  Not part of bench.py's contract yet.
usage: JAERO_TEST_AEROLC=1 python scripts/bench_aerol_c.py [channels] [steps]   (written without a GPU at hand)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jaero_amd import aerol_frames as AF  # noqa: E402
from jaero_amd import capi  # noqa: E402
from jaero_amd.demodulator import AeroLBank  # noqa: E402

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, nuniq = 3, 16
streams = []
for u in range(nuniq):
    frames, soft = AF.c_channel_case(9000 + u, K + W + 1, 20.0, inv=(bool(u & 1), bool(u & 2)), lead=0)
    streams.append(soft[: (K + W) * 4200])
uniq = torch.from_numpy(np.stack(streams)).cuda()                      # [nuniq, (K+W)*4200]
soft = uniq[torch.arange(nch, device="cuda") % nuniq].contiguous()     # [nch, (K+W)*4200]
counts = torch.full((nch,), 4200, dtype=torch.int32, device="cuda")
bank = AeroLBank(nch, 8400, max_softbits_per_write=4200, su_capacity=3 * (K + W) + 8)
st = torch.cuda.current_stream().cuda_stream
step = lambda i: bank.write_device(soft[:, i * 4200:(i + 1) * 4200].contiguous().data_ptr(), counts.data_ptr(), 4200, 4200, st)
for i in range(W):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(W, W + K):
    step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fn, voice = bank.read_voice(0)
sus = bank.read_sus(0)
print(json.dumps({"metric": "C-channel frames/s through the Aero-L bit pipeline", "value": round(K * nch / dt, 1), "ms_per_step": round(dt / K * 1e3, 3),
                  "channels": nch, "steps": K, "softbits_per_s": round(K * nch * 4200 / dt / 1e6, 2),
                  "frames_out_channel0": int(len(fn)), "crc_clean_units_channel0": int(sus[:, 14].sum()) if len(sus) else 0}))
