"""CPU tests: the product's host-side C (csrc/rd_post.c with its lock-free-reading helper pool, rd_helper.c, rd_synth.c) and the oracle under ThreadSanitizer and under
AddressSanitizer + UndefinedBehaviorSanitizer.  tests/native/host_stress.c is the driver: the post-process with 0..7 helper threads must return
the bytes of the caller's thread alone; 1000 cycles of configure / arm / shut down while two caller threads contend for the pool, every job exactly once; a fork with armed
helpers.  A sanitizer report ends the process with a non-zero exit code (halt_on_error)."""
import os
import subprocess

import pytest

from tests import helpers

NATIVE = os.path.join(helpers.ROOT, "tests", "native")


def _build(target):
    p = subprocess.run(["make", "-C", NATIVE, target], capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        if "sanitize" in p.stderr and ("cannot find" in p.stderr or "unrecognized" in p.stderr):
            pytest.skip("this compiler has no %s runtime: %s" % (target, p.stderr[-200:]))
        raise AssertionError(p.stderr[-2000:])
    return os.path.join(NATIVE, "build", "host_stress_" + target)


@pytest.mark.parametrize("target,env", [("tsan", {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1 die_after_fork=0"}),
                                        ("asan", {"ASAN_OPTIONS": "halt_on_error=1 detect_leaks=1", "UBSAN_OPTIONS": "halt_on_error=1 print_stacktrace=1"})])
def test_host_side_under_the_sanitizers(target, env):
    exe = _build(target)
    p = subprocess.run([exe, "1000"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "host_stress: ok" in p.stdout, (p.stdout + p.stderr)[-3000:]
    assert "Sanitizer" not in p.stderr, p.stderr[-3000:]
