"""Shared helpers for the tests (test infrastructure)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def relerr(a, b, floor=1e-300):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def seeded_lambda(sampler, seed, K, V):
    """`srand(seed); sampleGamma(K, V, 100) / 100` -- how LDA::LDA draws lambda
    (lda.cpp:71); `sampler` is any object with seed() and sample_gamma()."""
    sampler.seed(int(seed))
    return sampler.sample_gamma(int(K), int(V), 100) / 100.


def seeded_gamma(sampler, seed, K, B):
    sampler.seed(int(seed))
    return sampler.sample_gamma(int(K), int(B), 100) / 100.


class HipSampler(object):
    """seed()/sample_gamma() through the product's C ABI (host-side, no GPU needed)."""

    def __init__(self, lib):
        self.lib = lib

    def seed(self, s):
        self.lib.trlda_seed(int(s))

    def sample_gamma(self, m, n, k):
        out = np.zeros((m, n), order="F")
        self.lib.trlda_sample_gamma(m, n, k, out)
        return out


# parity bars (BASELINE.json: gamma / lambda within 1e-5 relative, fp64).  The tests hold the
# implementation to a much tighter figure; both are written here once.
NORTH_STAR_RTOL = 1e-5
TIGHT_RTOL = 1e-9


def kernel_resources(lib_path):
    """{mangled kernel name: {field: int}} from the AMDGPU metadata of the gfx950 code object inside
    a built library (the offload bundle's entry, read with llvm-readelf --notes): VGPRs, spills,
    scratch (`private_segment_fixed_size`) and LDS as the compiler allocated them."""
    import re
    import struct
    import subprocess
    import tempfile
    blob = open(lib_path, "rb").read()
    at = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert at >= 0, "no offload bundle in %s" % lib_path
    n, = struct.unpack_from("<Q", blob, at + 24)
    p, elf = at + 32, None
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, p)
        p += 24
        triple = blob[p:p + tl].decode()
        p += tl
        if "gfx950" in triple:
            elf = blob[at + off:at + off + size]
    assert elf, "no gfx950 code object"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(elf)
        f.flush()
        notes = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        fields = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\n", block)}
        out[name.group(1)] = fields
    return out
