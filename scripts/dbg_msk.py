import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jaero_amd import signalgen as G
from jaero_amd.demodulator import DemodulatorBank, MskSettings
nch, nsamp, chunk = 65, 30000, 4096
pcm, _, _ = G.channel_bank("msk", nch, nsamp, ebno_db=11.0, seed0=G.SEED_BASE + 100)
bank = DemodulatorBank([MskSettings(fb=1200, lockingbw=1800) for _ in range(nch)], ebno=True, status_log=True, capture_symbols=True, max_write_samples=chunk, softbit_capacity=nsamp)
for s in range(0, nsamp, chunk):
    bank.write(pcm[:, s:s+chunk])
for c in [0, 1, 31, 63, 64]:
    st = bank.read_status(c)
    print(c, "soft", len(bank.read_softbits(c)), "sym", len(bank.read_symbols(c)), "log", len(bank.read_status_log(c)), "nest", st.n_estimates, "mse", st.mse, st.freq_est)
