"""Whole-call time of update_parameters (the update loops end to end, inputs resident as a
DeviceBatch) at BASELINE.json's configurations: the fused device path against the plain launch
sequence, with and without the host-side gamma0 draw.

Every OnlineLDA call gets a mini-batch of its own (a model fed the same mini-batch again and
again fits it, and its E-steps then leave through the convergence test after a few iterations:
0.33 instead of 0.49 ms per call at K = 100 -- what rounds 1-2 and the first half of round 3
reported).  BatchLDA, whose epochs do revisit their corpus, starts every timed call from the same
lambda.

    python tools/update_rate.py [--configs small,c5a,c5b,c4] [--package trlda_amd]

Run on the GPU box from the repo root.  `--package` lets the same script time another tree
(e.g. an export of an earlier round under _r01/) for before/after figures.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    # name: (K, V, B, kind, calls)
    "small": (100, 7000, 200, "online", 40),
    "c3": (100, 7000, 1600, "online", 20),
    "c5a": (500, 100000, 512, "online", 6),
    "c5b": (500, 100000, 4096, "online", 8),
    "c4": (200, 50000, 12500, "batch", 7),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="small,c5a,c5b,c4")
    ap.add_argument("--root", default=ROOT, help="tree to import trlda_amd from")
    ap.add_argument("--modes", default="fused,plain")
    ap.add_argument("--reps", type=int, default=5, help="timed samples per row (median and max printed)")
    ap.add_argument("--samples", action="store_true", help="print every sample")
    ap.add_argument("--no-draw-ahead", action="store_true",
                    help="every gamma0 drawn in its turn on the model's stream (trlda_model_set_draw_ahead(0))")
    ap.add_argument("--host-draw", action="store_true",
                    help="draw gamma0 on the host (bit-exact glibc logarithms) instead of on the device")
    args = ap.parse_args()
    sys.path.insert(0, args.root)
    pkg = importlib.import_module("trlda_amd")
    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.models import BatchLDA, OnlineLDA
    from trlda_amd.utils.synthetic import make_corpus
    L = _ffi.lib()
    print("tree:", os.path.dirname(pkg.__file__))
    has_switch = hasattr(L, "trlda_model_set_fused_update")

    def sync(m):
        _ffi.check(L.trlda_model_synchronize(m._handle))

    for name in args.configs.split(","):
        K, V, B, kind, calls = CONFIGS[name]
        rng = np.random.RandomState(1)
        lam = np.asfortranarray(rng.gamma(100., .01, (K, V)))
        n_batches = calls + 1 if kind == "online" else 1
        corpora = [CSRDocuments(*make_corpus(B, V, seed=20150706 + K + 1000 * i, mean_unique=100))
                   for i in range(n_batches)]
        # time of the host-side gamma0 draw alone (sampleGamma(K, B, 100) / 100, lda.cpp:135)
        g = np.empty((K, B), order="F")
        L.trlda_sample_gamma_init(K, B, g)
        t = time.perf_counter()
        for _ in range(3):
            L.trlda_sample_gamma_init(K, B, g)
        draw_ms = (time.perf_counter() - t) / 3 * 1e3
        for mode in args.modes.split(","):
            if mode == "plain" and not has_switch:
                continue
            cls = OnlineLDA if kind == "online" else BatchLDA
            m = cls.__new__(cls)
            if kind == "online":
                m._num_documents, m._update_count = 1000000, 0
                m._ada_tau, m._ada_rho, m._ada_sq_norm = 1000., 1e-3, 1.
                m._ada_gradient = None
            m._setup(V, K, .1, .3, None, _lambda=lam)
            if has_switch:
                L.trlda_model_set_fused_update(m._handle, int(mode != "plain"))
                L.trlda_model_set_carry_rowsums(m._handle, int(mode != "plain"))
            if mode == "fused_sep":        # fused M-step, but every E-step launches its preamble
                if not hasattr(L, "trlda_model_set_next_preamble"):
                    continue
                L.trlda_model_set_next_preamble(m._handle, 0)
            if args.no_draw_ahead and hasattr(L, "trlda_model_set_draw_ahead"):
                L.trlda_model_set_draw_ahead(m._handle, 0)
            if args.host_draw and hasattr(L, "trlda_model_set_host_gamma_draw"):
                L.trlda_model_set_host_gamma_draw(m._handle, 1)
            batches = [m.upload(d) for d in corpora]
            variants = [("max_iter_tr=10", dict(max_iter_tr=10, max_iter_inference=20)),
                        ("max_iter_tr=0", dict(max_iter_tr=0, max_iter_inference=20))] \
                if kind == "online" else \
                [("1 epoch, max_iter_inference=100", dict(max_epochs=1, max_iter_inference=100)),
                 ("1 epoch, max_iter_inference=20", dict(max_epochs=1, max_iter_inference=20))]
            for label, kw in variants:
                if kind == "online":
                    m.lambdas = lam
                    m.update_parameters(batches[0], **kw)     # warm-up (allocations, code objects)
                    sync(m)
                    # (a stream of calls: the host runs ahead of the device, so single calls cannot
                    # be told apart -- `reps` whole streams instead, median and slowest)
                    samples = []
                    for r in range(args.reps):
                        m.lambdas = lam
                        sync(m)
                        t = time.perf_counter()
                        for i in range(calls):
                            m.update_parameters(batches[1 + i], **kw)
                        sync(m)
                        samples.append((time.perf_counter() - t) / calls)
                else:
                    m.update_parameters(batches[0], **kw)     # warm-up
                    samples = []
                    for i in range(max(calls, args.reps)):
                        m.lambdas = lam
                        sync(m)
                        t = time.perf_counter()
                        m.update_parameters(batches[0], **kw)
                        sync(m)
                        samples.append(time.perf_counter() - t)
                dt = float(np.median(samples))
                print("%-5s K=%d V=%d B=%d %-6s %-32s %9.3f ms/call (median of %d; max %.3f)  %10.0f docs/s  "
                      "(sampleGamma(K, B, 100) on the host alone: %.3f ms)"
                      % (name, K, V, B, mode, label, dt * 1e3, len(samples), max(samples) * 1e3, B / dt,
                         draw_ms))
                if args.samples:
                    print("      samples (ms):", " ".join("%.3f" % (x * 1e3) for x in samples))
            m.close()


if __name__ == "__main__":
    main()
