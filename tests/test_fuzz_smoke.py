"""CPU: a few rounds of every fuzzer under tests/fuzz/ with fixed seeds, so that they stay runnable and their cases stay in the suite: the Aero-L
kernels' device code on the host against the oracle (tests/fuzz/fuzz_aerol{b,c,p}_emul.py) and -- where oracle/_ref can run -- the oracle against the
unmodified reference (tests/fuzz/fuzz_oracle_vs_ref_{demod,burst,aerol}.py).  The long runs are recorded in profiles/r5_emul_fuzz.md."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, rounds, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", script), str(rounds), str(seed)], capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-2000:]
    assert p.returncode == 0 and " 0 mismatches" in p.stdout, tail


@pytest.mark.parametrize("script,rounds", [("fuzz_aerolb_emul.py", 6), ("fuzz_aerolc_emul.py", 4), ("fuzz_aerolp_emul.py", 5)])
def test_device_code_on_host_vs_oracle(oracle_mod, script, rounds):
    oracle_mod.lib()
    run(script, rounds, 7)


@pytest.mark.parametrize("script,rounds", [("fuzz_oracle_vs_ref_demod.py", 3), ("fuzz_oracle_vs_ref_burst.py", 2), ("fuzz_oracle_vs_ref_aerol.py", 6)])
def test_oracle_vs_unmodified_reference(oracle_mod, script, rounds):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref/jaero_ref not available here")
    try:
        oracle_mod.run_ref("msk", __import__("numpy").zeros(16, "int16"))
    except Exception as e:
        pytest.skip(f"_ref cannot run here: {e}")
    run(script, rounds, 7)
