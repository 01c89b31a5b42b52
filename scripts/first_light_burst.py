#!/usr/bin/env python3
"""First-light check of the burst HIP path against the C oracle (run on the GPU box; debugging aid, not a test)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jaero_amd import capi, signalgen as G  # noqa: E402
from jaero_amd.demodulator import BurstMskSettings, BurstOqpskSettings, DemodulatorBank  # noqa: E402
from oracle import oracle as O  # noqa: E402


def compare(tag, bank, c, ref, chunk):
    soft = bank.read_softbits(c)
    ev = bank.read_events(c)
    sym = bank.read_symbols(c)
    ok = True
    r_ev, g_ev = ref["events"], ev[np.lexsort((ev[:, 1], ev[:, 0]))] if len(ev) else ev
    r_ev = r_ev[np.lexsort((r_ev[:, 1], r_ev[:, 0]))]
    print(f"[{tag} ch{c}] soft {len(soft)} vs {len(ref['soft'])}  events {len(ev)} vs {len(r_ev)}  symbols {len(sym)} vs {len(ref['symbols'])}")
    if len(g_ev) != len(r_ev) or not np.array_equal(g_ev[:, :2], r_ev[:, :2]):
        ok = False
        print("  EVENT MISMATCH")
        n = max(len(g_ev), len(r_ev))
        for k in range(min(n, 40)):
            a = g_ev[k] if k < len(g_ev) else None
            b = r_ev[k] if k < len(r_ev) else None
            print("   ", a, b)
    else:
        d = np.abs(g_ev[:, 2] - r_ev[:, 2])
        rel = d / np.maximum(1.0, np.abs(r_ev[:, 2]))
        print("  event values max rel diff", rel.max() if len(rel) else 0)
        if len(rel) and rel.max() > 1e-6:
            ok = False
            for k in np.nonzero(rel > 1e-6)[0][:10]:
                print("   ", g_ev[k], r_ev[k])
    n = min(len(sym), len(ref["symbols"]))
    if n:
        d = np.abs(sym[:n] - ref["symbols"][:n])
        print("  symbols max abs diff", d.max(), "first bad", int(np.argmax(d.max(axis=1) > 1e-5)) if d.max() > 1e-5 else -1)
        ok &= d.max() < 1e-5 and len(sym) == len(ref["symbols"])
    n = min(len(soft), len(ref["soft"]))
    if n:
        hd = np.array_equal(soft[:n] >= 128, ref["soft"][:n] >= 128) and np.array_equal(soft[:n] == -1, ref["soft"][:n] == -1)
        md = np.max(np.abs(soft[:n].astype(int) - ref["soft"][:n].astype(int)))
        print("  hard decisions equal:", hd, " max soft byte diff", md)
        ok &= hd and md <= 1
    ok &= len(soft) == len(ref["soft"])
    print("  ->", "OK" if ok else "FAIL")
    return ok


def main():
    capi.lib()
    allok = True
    chunk = 4096
    # ---- burst OQPSK, synthetic ----
    nch, n = 3, 130000
    pcm = np.zeros((nch, n), np.int16)
    starts = [[20000, 76000], [15000, 70000], [30000, 90000]]
    for c in range(nch):
        pcm[c], _ = G.burst_oqpsk(n, burst_starts=starts[c], ndata_sym=1000, fc=8000 + 37.5 * (c - 1), ebno_db=15, seed=G.SEED_BASE + 40 + c)
    bank = DemodulatorBank(BurstOqpskSettings(), nch, device=0, capture_symbols=True, trace=True, max_write_samples=chunk, softbit_capacity=60000)
    for s in range(0, n, chunk):
        bank.write(pcm[:, s:s + chunk])
    for c in range(nch):
        ref = O.run_burst(O.burst_oqpsk_settings(), pcm[c], chunk=chunk, capture_symbols=True, trace=True)
        allok &= compare("burst-oqpsk", bank, c, ref, chunk)
    for w, nm in enumerate(["demod", "trident", "push", "hilbert", "front"]):
        pass
    bank.close()
    # ---- burst MSK 1200 on the recorded excerpt (if the fixture exists) ----
    fx = os.path.join(ROOT, "tests", "golden", "burst_msk_1200_sample1_excerpt.npz")
    if os.path.exists(fx):
        z = np.load(fx)
        x = z["pcm"]
        bank = DemodulatorBank(BurstMskSettings(), 1, device=0, capture_symbols=True, trace=True, max_write_samples=chunk, softbit_capacity=60000)
        for s in range(0, len(x), chunk):
            bank.write(x[None, s:s + chunk])
        ref = O.run_burst(O.burst_msk_settings(), x, chunk=chunk, capture_symbols=True, trace=True)
        allok &= compare("burst-msk-1200", bank, 0, ref, chunk)
        bank.close()
    print("ALL OK" if allok else "SOME FAILED")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
