"""
CPU sanitizer builds (no GPU sanitizer exists on this pool): the plain-C oracle under ASan + UBSan
through its own pytest suite, and -- opt-in, it recompiles the whole engine (~1 min) -- the HOST side of the HIP engine
(argument validation, arena layout arithmetic) under ASan + UBSan through tests/test_abi_and_host.py.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(target_dir):
    env = {k: v for k, v in os.environ.items() if k not in ("IPP_ORACLE_LIB", "IPP_HIP_LIB", "LD_PRELOAD")}
    return subprocess.run(["make", "-C", os.path.join(ROOT, target_dir), "asan"], capture_output=True, text=True, env=env, timeout=900)


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_under_address_and_ub_sanitizer():
    out = _run("oracle")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "ERROR: AddressSanitizer" not in out.stdout + out.stderr


@pytest.mark.skipif(os.environ.get("IPP_ASAN_ENGINE") != "1", reason="opt-in (recompiles the engine): IPP_ASAN_ENGINE=1 or `make -C ipp-rl_amd/csrc asan`")
def test_engine_host_side_under_address_and_ub_sanitizer():
    out = _run(os.path.join("ipp-rl_amd", "csrc"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "ERROR: AddressSanitizer" not in out.stdout + out.stderr
