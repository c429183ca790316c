"""SURVEY.md section 5 (race detection / sanitizers): the CPU oracle and the product's host-side structure builder
under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`).  GPU sanitizers are not available on
this pool, so the device code is covered by the parity tests instead."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_structure_builder_are_clean_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", "asan_driver")], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 failures" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
