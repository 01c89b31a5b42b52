"""GPU diagnostic: mixed write sizes with / without digital silence, both forms of the prefilter: symbol deviations from the oracle."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from jaero_amd import demodulator as B
from jaero_amd import signalgen as G
from oracle import oracle as O
from conftest import bank_settings, oracle_settings
opts = {"fb": 8400.0, "lockingbw": 8400.0}
sizes = [700, 3100, 4096, 50, 2048]
for silence in (False, True):
    pcm, _ = G.oqpsk(90000, fb=8400.0, fc=7985.0, ebno_db=10.0, seed=G.SEED_BASE + 8411)
    pcm = pcm.copy()
    if silence: pcm[40000:55000] = 0
    ref = O.run_demod(oracle_settings(O, "oqpsk", opts), pcm, chunk=[sizes[i % 5] for i in range(200)], capture_symbols=True)
    for form in ("fft", "direct"):
        os.environ["JAERO_PRE8400"] = form
        bank = B.DemodulatorBank([bank_settings("oqpsk", opts)], ebno=True, status_log=True, capture_symbols=True, max_write_samples=4096, softbit_capacity=len(pcm))
        s = k = 0
        while s < len(pcm):
            m = min(sizes[k % 5], len(pcm) - s); bank.write(pcm[None, s:s + m]); s += m; k += 1
        sym = bank.read_symbols(0); log = bank.read_status_log(0)
        d = np.abs(sym - ref["symbols"]); rows = np.flatnonzero(d.max(axis=1) > 1e-5)
        dl = np.abs(log[:, 1:5] - ref["status"][:, 1:5]).max(axis=0)
        print(f"silence {silence} {form}: max {d.max():.3g} at {np.unravel_index(d.argmax(), d.shape)}, rows > 1e-5: {len(rows)} {rows[:8]}, cols {d.max(axis=0)}, status cols {dl}", flush=True)
        bank.close()
