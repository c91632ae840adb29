"""The host-only code of libtrlda_hip.so (csrc/host_common.cpp, host_rng.cpp, text_docs.cpp,
eb_steps.cpp: the thread pool, the libc-compatible generator with its jump-ahead cache, the mmap
text parser, the empirical-Bayes Newton steps) built with a plain C++ compiler under
AddressSanitizer + UndefinedBehaviorSanitizer, and under ThreadSanitizer, and run through
tests/native/host_sanitize_main.cpp: multi-threaded parsing against the single-threaded result,
concurrent callers, 120 distinct draw shapes (threaded == serial bit for bit, the jump cache
filling up), refusals.  CPU only -- GPU sanitizers are not available on this pool."""
import subprocess

import pytest


@pytest.mark.parametrize("kind", ["address", "thread"])
@pytest.mark.timeout(600)
def test_host_code_under_sanitizers(kind):
    from trlda_amd import build
    exe = build.build_sanitized(kind)
    env = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1",
           "TSAN_OPTIONS": "halt_on_error=1", "PATH": "/usr/bin:/bin"}
    r = subprocess.run([exe] + (["threads"] if kind == "thread" else []), capture_output=True,
                       text=True, timeout=540, env=env)
    assert r.returncode == 0 and "HOST-SANITIZE-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: " not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
