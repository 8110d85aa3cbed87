"""The device source (host emulation) and the C oracle under AddressSanitizer + UndefinedBehaviorSanitizer.

Regression guard for an out-of-bounds LDS read found this way: with 8 vertices per foot the staging tile of
T = P [B A] is the LAST region of the 73 KB LDS map, and add_GtPG's fixed 10-column read batches ran past the end
of the allocation on the tile's last rows (harmless values, but outside the workgroup's LDS; with 4 vertices
the tile aliases dead stage vectors in the middle of the map, so the same reads stayed inside).  GPU
sanitizers are not available on the target pool, so this runs on the CPU build only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _san_lib(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_emulated_kernel_and_oracle_are_clean_under_asan_ubsan(tmp_path):
    asan, ubsan = _san_lib("libasan.so"), _san_lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    emu_so, ora_so = str(tmp_path / "libcmpc_emu_asan.so"), str(tmp_path / "libcmpc_oracle_asan.so")
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fPIC", "-shared"]
    subprocess.check_call(["g++", "-std=c++17", "-pthread", "-mfma", "-ffp-contract=off"] + flags + ["-o", emu_so, os.path.join(ROOT, "tests", "emu", "cmpc_emu.cpp")])
    subprocess.check_call(["gcc"] + flags + ["-o", ora_so, os.path.join(ROOT, "oracle", "cmpc_oracle.c"), "-lm"])
    env = dict(os.environ, LD_PRELOAD=f"{asan}:{ubsan}", ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "run_sanitized.py"), emu_so, ora_so],
                       capture_output=True, text=True, env=env, timeout=900)
    log = r.stdout + r.stderr
    assert "AddressSanitizer" not in log and "runtime error" not in log, log[-4000:]
    assert r.returncode == 0 and "sanitized run ok" in r.stdout, log[-4000:]
