import sys, numpy as np, torch
sys.path.insert(0, '.')
from jaero_amd import signalgen as G, demodulator as B, capi
from oracle import oracle as O
nch, nsamp, chunk = 4096, 110000, 4096
dev = torch.device("cuda", 0)
pcm, car, off = G.burst_oqpsk_torch(nch, nsamp, dev, ndata_sym=1500, ebno_db=15.0, seed=G.SEED_BASE + 40960, max_offset_sym=600)
bank = B.DemodulatorBank(B.BurstOqpskSettings(), nch, capture_symbols=True, trace=True, max_write_samples=chunk, softbit_capacity=30000)
for s in range(0, nsamp, chunk):
    bank.write(pcm[s:s + chunk], layout=capi.PCM_FRAME_MAJOR)
for c in (2047, 4094, 0):
    x = pcm[:, c].cpu().numpy()
    print("channel", c, "burst offset (samples)", float(off[c]), "first nonzero pcm", int(np.nonzero(x)[0][0]))
    ref = O.run_burst(O.burst_oqpsk_settings(), x, chunk=chunk, capture_symbols=True, trace=True)
    sym = bank.read_symbols(c)
    rs = ref["symbols"]
    no = (np.abs(rs[:, :2]).max(axis=1) < 1e-3) & (np.abs(sym[:, :2]).max(axis=1) < 1e-3)
    idx = np.nonzero(no)[0]
    print(" noise-only rows:", len(idx), "first", idx[:5], "last", idx[-5:] if len(idx) else None)
    for i in idx[:6]:
        print("  row", i, "ref", rs[i], "gpu", sym[i])
    ev = ref["events"]
    print(" first events (oracle):", ev[:6].tolist())
    print(" first events (gpu):", bank.read_events(c)[:6].tolist())
