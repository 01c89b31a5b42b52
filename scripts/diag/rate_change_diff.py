"""Where a live rate change on the HIP side departs from the oracle (diagnosis; run on a GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import bank_settings, oracle_settings
from jaero_amd import demodulator as B, signalgen as G
from oracle import oracle as O
O.build()

def run(kind, o0, o1, nch, nsamp, set_at, chunk, fbgen, Fsgen, seed0):
    kw = dict(fb=float(fbgen)) if kind == "oqpsk" else dict(fb=float(fbgen), Fs=float(Fsgen))
    pcm, _, _ = G.channel_bank(kind, nch, nsamp, ebno_db=12.0, seed0=seed0, **kw)
    bank = B.DemodulatorBank([bank_settings(kind, o0) for _ in range(nch)], ebno=True, status_log=True, capture_symbols=True, max_write_samples=chunk, softbit_capacity=nsamp)
    for s in range(0, set_at, chunk): bank.write(pcm[:, s:min(s + chunk, set_at)])
    bank.set_settings(bank_settings(kind, o1), channel=-1)
    for s in range(set_at, nsamp, chunk): bank.write(pcm[:, s:min(s + chunk, nsamp)])
    for c in range(nch):
        ref = O.run_demod(oracle_settings(O, kind, o0), pcm[c], chunk=chunk, capture_symbols=True, set_at=set_at, set_settings=oracle_settings(O, kind, o1))
        soft, sym, log = bank.read_softbits(c), bank.read_symbols(c), bank.read_status_log(c)
        n = min(len(soft), len(ref["soft"]))
        d = np.abs(soft[:n].astype(int) - ref["soft"][:n].astype(int))
        bad = np.nonzero(d > 1)[0]
        print(f"ch {c}: soft {len(soft)} vs {len(ref['soft'])}+{ref['pending']}, |diff|>1 at {len(bad)} places, first {bad[:8]}, last {bad[-3:]}, max {d.max()}")
        m = min(len(sym), len(ref["symbols"]))
        ds = np.abs(sym[:m] - ref["symbols"][:m]).max(axis=1)
        bs = np.nonzero(ds > 1e-5)[0]
        print(f"      symbols {sym.shape} vs {ref['symbols'].shape}: > 1e-5 at {len(bs)} places, first {bs[:6]}, max {ds.max():.3g}; first bad rows:")
        for i in bs[:3]: print("       ", i, sym[i], ref["symbols"][i])
        k = min(len(log), len(ref["status"]))
        dl = np.abs(log[:k, 1:4] - ref["status"][:k, 1:4]).max(axis=1)
        print(f"      status rows {log.shape} vs {ref['status'].shape}: max diff {dl.max():.3g} at row {dl.argmax()}")
    bank.close()

which = sys.argv[1]
if which == "oqpsk":
    run("oqpsk", {"fb": 10500.0, "lockingbw": 10500.0}, {"fb": 8400.0, "lockingbw": 8400.0, "freq_center": 8005.0}, 2, 90000, 20480, 4096, 8400, 48000, G.SEED_BASE + 7100 + 105)
else:
    run("msk", {"fb": 600.0, "lockingbw": 900.0, "Fs": 48000.0}, {"fb": 1200.0, "lockingbw": 1800.0, "Fs": 48000.0}, 3, 144000, 9000, 3000, 1200, 48000, G.SEED_BASE + 7700 + 6 + 4)
if which == "msk0":  # no setSettings at all: the same input through the 600 bps bank, several write sizes
    kind, o0 = "msk", {"fb": 600.0, "lockingbw": 900.0, "Fs": 48000.0}
    pcm, _, _ = G.channel_bank("msk", 3, 144000, ebno_db=12.0, seed0=G.SEED_BASE + 7700 + 6 + 4, fb=1200.0, Fs=48000.0)
    for chunk in (3000, 4096, 777):
        bank = B.DemodulatorBank([bank_settings(kind, o0) for _ in range(3)], ebno=True, status_log=True, capture_symbols=True, max_write_samples=chunk, softbit_capacity=144000)
        for s in range(0, 20000, chunk): bank.write(pcm[:, s:min(s + chunk, 20000)])
        for c in range(3):
            ref = O.run_demod(oracle_settings(O, kind, o0), pcm[c, :20000], chunk=chunk, capture_symbols=True)
            sym = bank.read_symbols(c)
            m = min(len(sym), len(ref["symbols"]))
            ds = np.abs(sym[:m] - ref["symbols"][:m]).max(axis=1)
            bs = np.nonzero(ds > 1e-5)[0]
            print(f"chunk {chunk} ch {c}: symbols {sym.shape} vs {ref['symbols'].shape}: > 1e-5 at {len(bs)} places, first {bs[:6]}, max {ds.max():.3g}")
        bank.close()
if which == "mskscan":
    kind = "msk"
    o0, o1 = {"fb": 600.0, "lockingbw": 900.0, "Fs": 48000.0}, {"fb": 1200.0, "lockingbw": 1800.0, "Fs": 48000.0}
    pcm, _, _ = G.channel_bank("msk", 3, 60000, ebno_db=12.0, seed0=G.SEED_BASE + 7700 + 6 + 4, fb=1200.0, Fs=48000.0)
    for set_at in (6000, 9000, 9040, 12000, 15000):
        chunk = 3000
        bank = B.DemodulatorBank([bank_settings(kind, o0) for _ in range(3)], ebno=True, status_log=True, capture_symbols=True, max_write_samples=chunk, softbit_capacity=60000)
        s = 0
        while s < 60000:
            if s == (set_at // 1000) * 1000 and set_at % 1000:
                bank.write(pcm[:, s:set_at]); s = set_at
            if s == set_at:
                pre = [len(bank.read_symbols(c)) for c in range(3)] if False else None
                bank.set_settings(bank_settings(kind, o1), channel=-1)
            m = min(chunk, 60000 - s)
            bank.write(pcm[:, s:s + m]); s += m
        for c in range(3):
            d = O.Demod(oracle_settings(O, kind, o0), capture_symbols=True)
            d.write(pcm[c, :set_at]); npre = len(d.take_symbols()); 
            ref = O.run_demod(oracle_settings(O, kind, o0), pcm[c], chunk=[3000] * (set_at // 3000) + ([set_at % 3000] if set_at % 3000 else []) + [3000] * 40, capture_symbols=True, set_at=set_at, set_settings=oracle_settings(O, kind, o1))
            sym = bank.read_symbols(c)
            m = min(len(sym), len(ref["symbols"]))
            ds = np.abs(sym[:m, :2] - ref["symbols"][:m, :2]).max(axis=1)
            bs = np.nonzero(ds > 1e-5)[0]
            print(f"set_at {set_at} ch {c}: pairs before the switch {npre}; re/im differ at {bs[:8]} (relative {bs[:8] - npre}), max {ds.max():.3g}")
        bank.close()
