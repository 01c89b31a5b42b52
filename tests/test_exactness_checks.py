"""CPU: scripts/atan2_check.c still compiles and passes in a short run (the full run is quoted in DESIGN 9 item 15): jaero_amd/csrc/jd_libm.h -- what
the sample kernels call since round 5 -- is a correctly rounded atan2 checked against __float128 and glibc 2.35's hypot restated, bit-identical to libm.  And the
multi-node device-sharing rule of jaero_amd/dist.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path, src, flags, libs):
    exe = str(tmp_path / (os.path.basename(src) + ".bin"))
    cc = ["g++", "-x", "c++"] if src.endswith("atan2_check.c") else ["gcc"]
    subprocess.check_call(cc + ["-O2", "-ffp-contract=off"] + flags + [os.path.join(ROOT, src), "-o", exe] + libs)
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_atan2_is_correctly_rounded_and_hypot_is_glibcs(tmp_path):
    exe = build(tmp_path, "scripts/atan2_check.c", ["-fopenmp", "-DHYPOT"], ["-lm", "-lquadmath"])
    out = subprocess.run([exe, "8", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert "hypot differences from libm: 0" in last, last
    # libm itself is NOT correctly rounded (about 1e-3 of its results): the check reports it, the function under test must not follow it
    assert any("differ from libm" in ln for ln in out.stdout.splitlines())


def test_ranks_share_a_device_is_a_job_wide_decision(monkeypatch):
    """The answer picks the backend, so every rank of a job must get the same one: it may depend on the launcher's local
    world size (torchrun, Slurm, Open MPI, MPICH) or on WORLD_SIZE, never on this rank's own LOCAL_RANK."""
    import torch

    from jaero_amd import dist as jd

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    for k in ("LOCAL_WORLD_SIZE", "SLURM_NTASKS_PER_NODE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS"):
        monkeypatch.delenv(k, raising=False)
    # srun / mpirun, 2 ranks on a 1-GPU node: both ranks must say "shared" (round 4: rank 0 said nccl, rank 1 gloo)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setenv("WORLD_SIZE", "2")
    answers = set()
    for lr in ("0", "1"):
        monkeypatch.setenv("LOCAL_RANK", lr)
        answers.add(jd.ranks_share_a_device())
    assert answers == {True}
    # Slurm says how many tasks a node runs: four nodes of eight on eight GPUs each share nothing, whatever this rank's index
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("WORLD_SIZE", "32")
    monkeypatch.setenv("SLURM_NTASKS_PER_NODE", "8(x4)")
    for lr in ("0", "5", "7"):
        monkeypatch.setenv("LOCAL_RANK", lr)
        assert jd.ranks_share_a_device() is False
    monkeypatch.setenv("SLURM_NTASKS_PER_NODE", "16(x2)")
    assert jd.ranks_share_a_device() is True
    monkeypatch.delenv("SLURM_NTASKS_PER_NODE")
    monkeypatch.setenv("OMPI_COMM_WORLD_LOCAL_SIZE", "8")
    assert jd.ranks_share_a_device() is False
    monkeypatch.delenv("OMPI_COMM_WORLD_LOCAL_SIZE")
    # nobody said: conservative and the same everywhere (32 ranks > 8 devices -> gloo unless JAERO_DIST_BACKEND says otherwise)
    for lr in ("0", "5"):
        monkeypatch.setenv("LOCAL_RANK", lr)
        assert jd.ranks_share_a_device() is True
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert jd.ranks_share_a_device() is False
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "16")
    assert jd.ranks_share_a_device() is True
