"""Build libtrlda_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m trlda_amd.build            # build if stale
    python -m trlda_amd.build --force
    python -m trlda_amd.build --sanitize address|thread   # host-only code under ASan+UBSan / TSan
    python -m trlda_amd.build --variant NAME -DX=1 ...    # libtrlda_hip.NAME.so with extra flags
                                                          # (use it with TRLDA_LIB=<path>)
"""
import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "libtrlda_hip.so")
# the HIP translation unit (kernels, launch sequences, the C ABI over them) and the host-only
# ones (no HIP: they also build with a plain C++ compiler under the sanitizers, `--sanitize`)
SOURCES = ["trlda_hip.hip", "host_common.cpp", "host_rng.cpp", "text_docs.cpp", "eb_steps.cpp",
           "batch_index.cpp"]
HOST_SOURCES = SOURCES[1:]


def _headers():
    """Every header the library is built from: csrc/*.h (found, not listed -- a new header
    cannot be forgotten; tests/test_boundary.py checks each one is #included somewhere) and
    the public C ABI."""
    import glob
    found = sorted(os.path.basename(h) for h in glob.glob(os.path.join(_CSRC, "*.h")))
    return found + [os.path.join("..", "..", "include", "trlda_hip.h")]


HEADERS = _headers()
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
               "-munsafe-fp-atomics", "-Wall"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def fastdocs_path():
    import sysconfig
    return os.path.join(_PKG, "_fastdocs" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_fastdocs(force=False, verbose=False):
    """Compile csrc/fastdocs.c (CPython extension: list-of-tuples <-> CSR) with the C compiler."""
    import sysconfig
    out, src = fastdocs_path(), os.path.join(_CSRC, "fastdocs.c")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("no C compiler for trlda_amd._fastdocs")
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-I" + sysconfig.get_paths()["include"],
           "-o", out, src]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False):
    """Compile the HIP kernels + C ABI into trlda_amd/libtrlda_hip.so."""
    build_fastdocs(force=force, verbose=verbose)
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(_CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=_CSRC)
    return LIB_PATH


def build_variant(name, extra_flags, verbose=False):
    """The same sources with extra compiler flags / defines as trlda_amd/libtrlda_hip.<name>.so, for
    the tuning sweeps and diagnostic builds under tools/ (run them with TRLDA_LIB=<that path>):
    the package's own library is never overwritten by an experiment."""
    out = os.path.join(_PKG, "libtrlda_hip.%s.so" % name)
    cmd = [_hipcc()] + HIPCC_FLAGS + list(extra_flags) + ["-o", out] + \
        [os.path.join(_CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=_CSRC)
    return out


def build_sanitized(kind="address", verbose=False):
    """The host-only translation units + tests/native/host_sanitize_main.cpp as ONE executable under
    a sanitizer (plain g++: no HIP in these files): `address` = ASan + UBSan, `thread` = TSan.
    GPU sanitizers are not available on this pool; this is the host side -- threads, mmap
    parsing, the jump-ahead cache, the K-sized numerics.  Returns the executable's path."""
    flags = {"address": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
             "thread": ["-fsanitize=thread"]}[kind]
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no C++ compiler for the sanitizer build")
    out_dir = os.path.join(os.path.dirname(_PKG), "build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "host_sanitize_" + kind)
    driver = os.path.join(os.path.dirname(_PKG), "tests", "native", "host_sanitize_main.cpp")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", "-Wall"] + flags + \
        ["-o", out, driver] + [os.path.join(_CSRC, f) for f in HOST_SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        kind = sys.argv[sys.argv.index("--sanitize") + 1] if len(sys.argv) > sys.argv.index("--sanitize") + 1 \
            else "address"
        exe = build_sanitized(kind, verbose=True)
        sys.exit(subprocess.run([exe] + (["threads"] if kind == "thread" else [])).returncode)
    if "--variant" in sys.argv:
        # python -m trlda_amd.build --variant NAME [flags for hipcc ...]
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
