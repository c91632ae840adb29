"""The boundary from plain C: tests/native/c_client.c, compiled with gcc against include/trlda_hip.h
and linked to libtrlda_hip.so -- no Python, torch or C++ in the client -- run on the GPU box; its
output is checked against the size-independent invariants of SURVEY.md a17."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client_of_the_boundary(hip_lib, tmp_path):
    from trlda_amd import _ffi
    assert _ffi.device_count() >= 1
    libdir = os.path.join(ROOT, "trlda_amd")
    exe = str(tmp_path / "c_client")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "native", "c_client.c"), "-o", exe,
                    "-L", libdir, "-l:libtrlda_hip.so", "-Wl,-rpath," + libdir, "-lm"], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    text = out.stdout.decode()
    m = re.search(r"estep counts (\S+) sstats (\S+) gamma (\S+) expect_gamma (\S+) iters (\d+) (\d+)", text)
    counts, ssum, gsum, gexp, itmin, itmax = [float(x) for x in m.groups()]
    assert abs(ssum - counts) < 1e-9 * counts
    assert abs(gsum - gexp) < 1e-9 * gexp
    assert 1 <= itmin <= itmax <= 20
    ups = re.findall(r"update (\d) rho (\S+) lambda (\S+) expect (\S+) count (\d+)", text)
    assert len(ups) == 2
    for i, (call, rho, lsum, expect, count) in enumerate(ups):
        assert float(rho) == (100. + i) ** -.7 or abs(float(rho) - (100. + i) ** -.7) < 1e-15
        assert abs(float(lsum) - float(expect)) < 1e-9 * float(expect)
        assert int(count) == i + 1
    assert re.search(r"null batch -> -\d+ \(.+\)", text)
