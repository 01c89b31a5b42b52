"""GPU (-m gpu): the multi-GPU edge operations of the C ABI (jaero_fan_out_pcm / jaero_gather_softbits).  This lease has one GPU, so a
communicator has one rank here: once without RCCL (local copies) and once as a one-rank RCCL communicator, whose operations go through
ncclSend / ncclRecv to itself inside a group -- the same calls a multi-rank communicator makes towards its peers (RCCL loaded by dlopen).
The rank arithmetic of several ranks is covered on the CPU (tests/test_capi_host.py, tests/test_dist_gloo.py)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("with_rccl", [False, True])
def test_edge_operations_one_rank(with_rccl):
    import torch

    from jaero_amd import capi

    L = capi.lib()
    dev = torch.device("cuda:0")
    comm = C.c_void_p()
    ident = (C.c_char * 128)()
    if with_rccl:
        capi.check(L.jaero_comm_get_unique_id(ident))
    capi.check(L.jaero_comm_create(0, 0, 1, ident if with_rccl else None, C.byref(comm)))
    try:
        nsamp, nch, cap = 257, 70, 96
        rng = np.random.default_rng(5)
        frames = torch.from_numpy(rng.integers(-32768, 32767, (nsamp, nch), dtype=np.int16)).to(dev)
        mine = torch.zeros((nsamp, nch), dtype=torch.int16, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        capi.check(L.jaero_fan_out_pcm(comm, 0, frames.data_ptr(), nsamp, nch, mine.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(mine, frames)
        soft = torch.from_numpy(rng.integers(0, 256, (nch, cap), dtype=np.int16)).to(dev)
        cnt = torch.from_numpy(rng.integers(0, cap, nch, dtype=np.int32)).to(dev)
        soft_all = torch.zeros_like(soft)
        cnt_all = torch.zeros_like(cnt)
        capi.check(L.jaero_gather_softbits(comm, 0, soft.data_ptr(), cnt.data_ptr(), nch, cap, soft_all.data_ptr(), cnt_all.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(soft_all, soft) and torch.equal(cnt_all, cnt)
        # a peer that does not exist is refused, not waited for
        with pytest.raises(capi.JaeroError):
            capi.check(L.jaero_fan_out_pcm(comm, 1, frames.data_ptr(), nsamp, nch, mine.data_ptr(), st))
    finally:
        L.jaero_comm_destroy(comm)


def test_fan_out_feeds_a_bank():
    """The slice a rank receives is what jaero_write takes as frame-major PCM: fan out, demodulate, gather -- against the oracle."""
    import torch

    from conftest import bank_settings, oracle_settings
    from jaero_amd import capi, demodulator as B, signalgen as G
    from oracle import oracle as O

    O.build()
    L = capi.lib()
    dev = torch.device("cuda:0")
    nch, nsamp, chunk = 3, 40960, 4096
    pcm, _, _ = G.channel_bank("oqpsk", nch, nsamp, ebno_db=12.0, seed0=G.SEED_BASE + 9100)
    comm = C.c_void_p()
    capi.check(L.jaero_comm_create(0, 0, 1, None, C.byref(comm)))
    bank = B.DemodulatorBank([bank_settings("oqpsk", {}) for _ in range(nch)], max_write_samples=chunk, softbit_capacity=nsamp)
    st = torch.cuda.current_stream().cuda_stream
    frames_all = torch.from_numpy(np.ascontiguousarray(pcm.T)).to(dev)          # [nsamp][nch]
    mine = torch.empty((chunk, nch), dtype=torch.int16, device=dev)
    for s in range(0, nsamp, chunk):
        capi.check(L.jaero_fan_out_pcm(comm, 0, frames_all[s:s + chunk].data_ptr(), chunk, nch, mine.data_ptr(), st))
        bank.write(mine, layout=capi.PCM_FRAME_MAJOR, stream=st)
    for c in range(nch):
        ref = O.run_demod(oracle_settings(O, "oqpsk", {}), pcm[c], chunk=chunk)
        got = bank.read_softbits(c)
        n = len(ref["soft"])
        assert len(got) == n + ref["pending"] and np.array_equal(got[:n] >= 128, ref["soft"] >= 128)
    bank.close()
    L.jaero_comm_destroy(comm)
