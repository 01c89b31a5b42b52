"""GPU diagnostic: which of (write size, AFC) makes the FFT form of the 8400 prefilter differ from the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from conftest import load_golden  # noqa: E402
from jaero_amd import demodulator as B  # noqa: E402
from jaero_amd.demodulator import OqpskSettings  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_parity import feed  # noqa: E402

g = load_golden("oqpsk_8400_afc_chunk1500_dcd")
pcm = g["pcm"].reshape(1, -1)[:, :60000]
for chunk in (1500, 2048, 2047, 1024, 3000, 4096):
    for afc in (0, 1):
        ref = O.run_demod(O.oqpsk_settings(lockingbw=8400.0, fb=8400.0), pcm[0], chunk=chunk, afc=bool(afc))
        for form in ("fft", "direct"):
            os.environ["JAERO_PRE8400"] = form
            st = OqpskSettings(freq_center=8000.0, lockingbw=8400.0, fb=8400.0, coarsefreqest_fft_power=14, signalthreshold=0.65)
            bank = B.DemodulatorBank(st, 1, ebno=True, status_log=True, max_write_samples=8192, softbit_capacity=pcm.shape[1])
            bank.set_flags(afc=bool(afc), cpu_reduce=False)
            feed(bank, pcm, chunk)
            soft, log = bank.read_softbits(0), bank.read_status_log(0)
            n = len(ref["soft"])
            bad = np.flatnonzero((soft[:n] >= 128) != (ref["soft"] >= 128))
            dd = np.abs(log[:, 1:4] - ref["status"][:, 1:4])
            print(f"chunk {chunk} afc {afc} {form}: hard mismatches {len(bad)} of {n}, first {bad[:3]}, status max d {dd.max(axis=0)}", flush=True)
            bank.close()
