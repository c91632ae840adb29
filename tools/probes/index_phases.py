"""Where the host time of the mini-batch index goes (csrc/batch_index.cpp): an instrumented copy of
the source (a clock read between the phases), compiled with g++ -O3 into /tmp and run on the headline
batch.  Host only: runs anywhere.

    python tools/probes/index_phases.py [--batch 200] [--words 7000]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "trlda_amd", "csrc")

MARKS = [
    ("    int max_n = 0, n_wg = 0, n_xrows = 0;\n    for (int d = 0; d < B; ++d) {\n        const int n = indptr[d + 1] - indptr[d];\n        if (indptr[d + 1] < indptr[d])\n            return fail(TRLDA_ERR_ARG, \"indptr must be non-decreasing\");\n        max_n = std::max(max_n, n);\n        const int c = segments_of(n);\n        n_wg += c;\n        n_xrows += c > 1 ? c : 0;\n    }\n    if (n_xrows == 0)",
     None, "START"),
    ("    std::vector<int32_t> &wptr = x->wptr;\n    wptr.assign((size_t)V + 1, 0);\n    {\n        int32_t *cnt", "lengths", None),
    ("    // ONE scan of the vocabulary: offsets", "zero + histogram + validation", None),
    ("    int over[kLevels];\n", "scan of the vocabulary", None),
    ("    x->total = off;\n    return TRLDA_OK;", "levels, layout", None),
    ("    // stable counting sort of the CSR positions by word id, and the words' count sums\n", "copies (csr, wptr)", None),
    ("    // documents by decreasing length, equal lengths in document order", "counting sort of the entries + count sums", None),
    ("    x->sorted_len.resize(Bz);", "documents by length", None),
    ("    x->split_pays = false;\n    if (n_wg > 0) {", "meta + padded id rows", None),
    ("        std::memcpy(active, awords.data(), (size_t)n_active * 4);", "split layout; second scan (flags, active, lengths)", None),
    ("        // descriptors for the merged launch: counting sort by length, longest first, the short\n", "very long lists", None),
]


def instrumented():
    s = open(os.path.join(CSRC, "batch_index.cpp")).read()
    s = s.replace("namespace trlda_host {\n\nnamespace {", """#include <chrono>
static double g_t[32]; static std::chrono::steady_clock::time_point g_t0;
#define MARK(n) do { auto t_ = std::chrono::steady_clock::now(); g_t[n] += std::chrono::duration<double, std::micro>(t_ - g_t0).count(); g_t0 = t_; } while (0)
extern "C" void idx_reset() { for (int i = 0; i < 32; ++i) g_t[i] = 0; }
extern "C" double idx_get(int i) { return g_t[i]; }
namespace trlda_host {

namespace {""", 1)
    names = []
    # only the plan's first statement (the second copy of that text is batch_index_check_lengths')
    plan_at = s.index("int batch_index_plan(")
    for anchor, name, special in MARKS:
        at = s.index(anchor, plan_at)
        if special == "START":
            ins = "    g_t0 = std::chrono::steady_clock::now();\n"
        else:
            ins = "    MARK(%d);\n" % len(names)
            names.append(name)
        s = s[:at] + ins + s[at:]
    tail = "            e[0] = w; e[1] = wptr[(size_t)w]; e[2] = wptr[(size_t)w + 1] - wptr[(size_t)w]; e[3] = 0;\n        }\n    }\n}\n"
    at = s.index(tail) + len(tail) - 2
    s = s[:at] + "    MARK(%d);\n" % len(names) + s[at:]
    names.append("descriptors")
    for h in ("batch_index.h", "host_common.h", "index_params.h"):
        s = s.replace('#include "%s"' % h, '#include "%s"' % os.path.join(CSRC, h))
    s = s.replace('#include "../../include/trlda_hip.h"', '#include "%s"' % os.path.join(ROOT, "include", "trlda_hip.h"))
    return s, names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--words", type=int, default=7000)
    ap.add_argument("--reps", type=int, default=3000)
    a = ap.parse_args()
    src, names = instrumented()
    open("/tmp/trlda_index_phases.cpp", "w").write(src)
    for cxx, flags in (("g++", []), ("g++", ["-march=native"])):
        so = "/tmp/libtrlda_index_phases%s.so" % ("_native" if flags else "")
        subprocess.run([cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread"] + flags +
                       ["-o", so, "/tmp/trlda_index_phases.cpp", os.path.join(CSRC, "host_common.cpp")], check=True)
        from trlda_amd.utils.synthetic import make_corpus
        lib = C.CDLL(so)
        ip, ii, cc = make_corpus(a.batch, a.words, seed=20150707, mean_unique=100)
        f = lib.trlda_debug_batch_index_rate
        f.restype = C.c_double
        lib.idx_get.restype = C.c_double
        p32 = lambda x: x.ctypes.data_as(C.POINTER(C.c_int32))
        f(a.words, a.batch, p32(ip), p32(ii), p32(cc), 256, 300, 1)
        lib.idx_reset()
        total = f(a.words, a.batch, p32(ip), p32(ii), p32(cc), 256, a.reps, 1)
        print("%s %s: %.1f us per batch of %d documents (%d entries, V = %d), clock reads included" % (
            cxx, " ".join(flags) or "(baseline x86-64)", total, a.batch, len(ii), a.words))
        for i, n in enumerate(names):
            print("    %-52s %6.2f us" % (n, lib.idx_get(i) / a.reps))


if __name__ == "__main__":
    main()
