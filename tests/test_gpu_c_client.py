"""The boundary from plain C: tests/native/c_client.c, compiled with gcc against include/trlda_hip.h
and linked to libtrlda_hip.so -- no Python, torch or C++ in the client -- run on the GPU box; its
output is checked against the size-independent invariants of SURVEY.md a17, and -- VERDICT r4 item
10 -- against VALUES: the client runs the compiled reference's golden vector f1a (flattened here
into one binary file) through the one-shot entry, through the handle API and, five times in a row, as
a stream through two lanes (trlda_model_estep_io_ahead on device arrays), and prints the largest
relative errors of gamma and the statistics."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import TIGHT_RTOL, golden

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
    f = golden("f1a_estep")
    K, V, B, max_iter = int(f["K"]), int(f["V"]), int(f["B"]), 20
    key = "it%d_thr0.001" % max_iter
    vec = str(tmp_path / "f1a.bin")
    with open(vec, "wb") as fh:
        fh.write(np.array([K, V, B, len(f["ids"]), int(f["lambda_seed"]), int(f["gamma0_seed"]), max_iter, 0],
                          dtype=np.int32).tobytes())
        for name in ("indptr", "ids", "cnts"):
            fh.write(np.ascontiguousarray(f[name], dtype=np.int32).tobytes())
        fh.write(np.ascontiguousarray(f["alpha"], dtype=np.float64).tobytes())
        for name in ("gamma_" + key, "sstats_" + key):          # column-major, as the C ABI
            fh.write(np.asfortranarray(f[name], dtype=np.float64).tobytes(order="F"))
        fh.write(np.ascontiguousarray(f["iters_" + key], dtype=np.int32).tobytes())
    out = subprocess.run([exe, vec], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    text = out.stdout.decode()
    lanes = re.search(r"golden lanes .* through (\d+) of (\d+)", text)
    assert lanes and lanes.group(1) == lanes.group(2), text       # every call of the stream went through a lane
    for api in ("oneshot", "handle", "lanes"):
        g = re.search(r"golden %s gamma_err (\S+) sstats_err (\S+) zeros_agree (\d) iters_equal (\d)" % api, text)
        assert g, text
        assert float(g.group(1)) < TIGHT_RTOL and float(g.group(2)) < TIGHT_RTOL, (api, g.groups())
        assert g.group(3) == "1" and g.group(4) == "1", (api, g.groups())
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
