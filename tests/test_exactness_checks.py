"""CPU: scripts/atan2_check.c still compiles and passes in a short run (the full run is quoted in DESIGN 9 item 15): scripts/ubench/jd_atan2.h -- round 4,
NOT in the product -- is a correctly rounded atan2 checked against __float128 and glibc 2.35's hypot restated, bit-identical to libm.  And the
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


def test_ranks_share_a_device_without_local_world_size(monkeypatch):
    """Launchers other than torchrun export WORLD_SIZE for all nodes and no LOCAL_WORLD_SIZE: device sharing is then decided from this
    rank's own LOCAL_RANK and the device count, not from the global world size."""
    import torch

    from jaero_amd import dist as jd

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "32")  # four nodes of eight
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert jd.ranks_share_a_device() is False
    monkeypatch.setenv("LOCAL_RANK", "9")
    assert jd.ranks_share_a_device() is True
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert jd.ranks_share_a_device() is False
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "16")
    assert jd.ranks_share_a_device() is True
