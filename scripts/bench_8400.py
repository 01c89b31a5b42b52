"""First measurement script for the 8400 bps C-channel demodulator (row f4): Msamples/s of a bank fed from HBM, like bench.py's
default workload but at fb = 8400 (prefilter + sample loop + centre-weighted coarse estimate).  Not part of bench.py's contract yet.
usage: python scripts/bench_8400.py [channels] [steps]   (written without a GPU at hand: run it, then fold it into bench.py)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jaero_amd import capi, signalgen  # noqa: E402
from jaero_amd.demodulator import DemodulatorBank, OqpskSettings  # noqa: E402

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, chunk = 4, 4096
dev = torch.device("cuda", 0)
nsamp = (K + W) * chunk
pcm, bits, _ = signalgen.oqpsk_torch(nch, nsamp, dev, fb=8400.0, ebno_db=12.0, seed=signalgen.SEED_BASE + 84)
bank = DemodulatorBank(OqpskSettings(fb=8400.0, lockingbw=8400.0), nch, ebno=True, max_write_samples=chunk, softbit_capacity=int(nsamp * 8400 / 48000) + 64)
st = torch.cuda.current_stream().cuda_stream
for i in range(W):
    bank.write(pcm[i * chunk:(i + 1) * chunk], layout=capi.PCM_FRAME_MAJOR, stream=st)
torch.cuda.synchronize()
bank.profile_enable(True)
t0 = time.perf_counter()
for i in range(W, W + K):
    bank.write(pcm[i * chunk:(i + 1) * chunk], layout=capi.PCM_FRAME_MAJOR, stream=st)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
samp_ms, samp_n = bank.profile_read(0)
coarse_ms, coarse_n = bank.profile_read(1)
status = [bank.read_status(c) for c in range(min(8, nch))]
print(json.dumps({"metric": "Msamples/s of real 48 kHz PCM through the 8400 bps OQPSK (C channel) demodulator", "value": round(K * chunk * nch / dt / 1e6, 2),
                  "ms_per_step": round(dt / K * 1e3, 3), "channels": nch, "steps": K,
                  "kernel_ms_per_step": {"sample_loop": round(samp_ms / K, 3), "coarse": round(coarse_ms / K, 3),
                                         "prefilter_and_rest": round(dt / K * 1e3 - samp_ms / K - coarse_ms / K, 3)},
                  "locked_of_checked": int(sum(int(s.signal) for s in status))}))
