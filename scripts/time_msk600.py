"""Timing of the 600 bps MSK sample kernel (78 + 82 split of the filter history, two wavefronts per CU).
usage: python scripts/time_msk600.py [channels]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jaero_amd import capi  # noqa: E402
from jaero_amd.demodulator import DemodulatorBank, MskSettings  # noqa: E402

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
chunk, K = 4096, 6
rng = np.random.default_rng(1)
pcm = torch.from_numpy(rng.integers(-3000, 3000, size=(chunk, 64), dtype=np.int16)).cuda().repeat(1, nch // 64).contiguous()
bank = DemodulatorBank(MskSettings(fb=600.0, lockingbw=900.0, freq_center=1000.0), nch, ebno=True, max_write_samples=chunk, softbit_capacity=4096)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    bank.write(pcm, layout=capi.PCM_FRAME_MAJOR, stream=st)
torch.cuda.synchronize()
bank.profile_enable(True)
t0 = time.perf_counter()
for _ in range(K):
    bank.write(pcm, layout=capi.PCM_FRAME_MAJOR, stream=st)
    bank.discard_softbits(st)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, n = bank.profile_read(0)
print({"channels": nch, "Msamples_s": round(K * chunk * nch / dt / 1e6, 1), "sample_loop_ms_per_launch": round(ms / max(n, 1), 3)})
