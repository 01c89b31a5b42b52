"""First-light check on a GPU box: HIP path vs C oracle on a few synthetic channels (prints diffs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from jaero_amd import signalgen as G
from jaero_amd.demodulator import DemodulatorBank, OqpskSettings, MskSettings
from jaero_amd import capi

def run(kind, nch, nsamp, chunk):
    if kind == "oqpsk":
        pcm, carriers, _ = G.channel_bank("oqpsk", nch, nsamp, ebno_db=10.0)
        setts = [OqpskSettings() for _ in range(nch)]
        osett = [O.oqpsk_settings() for _ in range(nch)]
    else:
        pcm, carriers, _ = G.channel_bank("msk", nch, nsamp, ebno_db=12.0)
        setts = [MskSettings(fb=1200, lockingbw=1800) for _ in range(nch)]
        osett = [O.msk_settings() for _ in range(nch)]
    t = time.time()
    bank = DemodulatorBank(setts, device=0, ebno=True, status_log=True, capture_symbols=True, max_write_samples=chunk, softbit_capacity=int(nsamp * 0.25) + 64)
    for s in range(0, nsamp, chunk):
        bank.write(pcm[:, s:s + chunk])
    outs = [(bank.read_softbits(c), bank.read_status_log(c), bank.read_symbols(c)) for c in range(nch)]
    print(kind, "gpu time", time.time() - t)
    for c in range(nch):
        o = O.run_demod(osett[c], pcm[c], chunk=chunk, capture_symbols=True)
        gs, gl, gy = outs[c]
        n = min(len(gs), len(o["soft"]))
        hard_eq = np.array_equal(gs[:n] >= 128, o["soft"][:n] >= 128) if n else True
        maxd = int(np.max(np.abs(gs[:n].astype(int) - o["soft"][:n].astype(int)))) if n else 0
        ns = min(len(gy), len(o["symbols"]))
        symd = float(np.max(np.abs(gy[:ns] - o["symbols"][:ns]))) if ns else 0.0
        nl = min(len(gl), len(o["status"]))
        std = float(np.max(np.abs(gl[:nl, :4] - o["status"][:nl, :4]))) if nl else 0.0
        ebd = float(np.max(np.abs(gl[:nl, 4] - o["status"][:nl, 4]))) if nl else 0.0
        print(f" ch{c} fc={carriers[c]:.1f} soft {len(gs)} vs {len(o['soft'])} hard_eq={hard_eq} maxsoftdiff={maxd} "
              f"sym {len(gy)} vs {len(o['symbols'])} maxsymdiff={symd:.3e} status {len(gl)} vs {len(o['status'])} maxdiff={std:.3e} ebno diff={ebd:.3e} mse={o['mse']:.4f}")
        if not hard_eq or len(gs) != len(o["soft"]):
            k = np.nonzero(gs[:n] != o["soft"][:n])[0][:10]
            print("   first diffs at", k, gs[k], o["soft"][k])
            if ns:
                d = np.abs(gy[:ns] - o["symbols"][:ns]).max(axis=1)
                bad = np.nonzero(d > 1e-6)[0][:5]
                print("   first symbol diffs at", bad, gy[bad], o["symbols"][bad])
            print("   status head gpu", gl[:3], "oracle", o["status"][:3])
    bank.close()

def viterbi():
    rng = np.random.default_rng(5)
    L = capi.lib()
    nblk, nsoft = 6, 5078
    msg = rng.integers(0, 256, size=(nblk, (nsoft // 2 - 8) // 8), dtype=np.uint8)
    soft = np.zeros((nblk, nsoft), np.uint8)
    for b in range(nblk):
        coded = O.encode_bits(msg[b])[:nsoft]
        x = (coded.astype(float) * 2 - 1) + rng.normal(0, 0.5, coded.shape)
        soft[b, :len(coded)] = np.clip(np.round(x * 64 + 128), 0, 255).astype(np.uint8)
        soft[b, len(coded):] = 128
    out = np.zeros((nblk, nsoft // 2), np.uint8)
    capi.check(L.jaero_viterbi_decode_soft(0, soft.ctypes.data, nblk, nsoft, out.ctypes.data, 0, None))
    ok = True
    for b in range(nblk):
        ref = O.Codec().decode_soft(soft[b])
        ok &= np.array_equal(ref[: nsoft // 2 - 6], out[b, : nsoft // 2 - 6])
    print("viterbi decode_soft equal:", ok)
    # continuous
    ov = np.zeros((nblk, 64), np.uint8)
    codecs = [O.Codec(24) for _ in range(nblk)]
    allok = True
    for it in range(3):
        soft = rng.integers(0, 256, size=(nblk, nsoft), dtype=np.uint8)
        outc = np.zeros((nblk, nsoft // 2), np.uint8)
        nb = np.zeros(nblk, np.int32)
        capi.check(L.jaero_viterbi_continuous(0, soft.ctypes.data, nblk, nsoft, 24, ov.ctypes.data, outc.ctypes.data, nb.ctypes.data, 0, None))
        for b in range(nblk):
            ref = codecs[b].decode_continuous(soft[b])
            good = len(ref) == nb[b] and np.array_equal(ref[: nb[b] - 8], outc[b, : nb[b] - 8])
            if not good: print("  cont mismatch it", it, "blk", b, len(ref), nb[b], np.nonzero(ref[:nb[b]] != outc[b,:nb[b]])[0][:10])
            allok &= good
    print("viterbi continuous equal:", allok)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "viterbi"): viterbi()
    if which in ("all", "oqpsk"): run("oqpsk", 5, 48000 * 2, 4096)
    if which in ("all", "msk"): run("msk", 3, 48000 * 2, 5000)
