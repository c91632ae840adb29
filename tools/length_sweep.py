"""How long one E-step takes as a function of the documents' lengths: the cliffs between the
document-kernel variants (estep_kernels.h 3c, estep_wide.h).

    python tools/length_sweep.py [--topics 100 --words 7000 --batch 200]

Two series, both on B-document batches at max_iter 20, threshold 0 (fixed work):
  all   every document has exactly n unique words
  one   B - 1 documents of 100 words and ONE of n words (a launch lasts as long as its longest
        document: what a single long document costs the batch)
Run on the GPU box from the repo root.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topics", type=int, default=100)
    ap.add_argument("--words", type=int, default=7000)
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--lengths", default="32,64,100,128,129,144,145,160,192,193,224,256,320,400,600")
    ap.add_argument("--series", default="all,one")
    ap.add_argument("--wide", action="store_true", help="force the single-orientation kernel")
    ap.add_argument("--no-deferred", action="store_true",
                    help="every step launches its statistics kernel (rounds 1-4) instead of leaving them "
                         "to the next step's document launch (trlda_model_set_deferred_stats)")
    ap.add_argument("--lanes", type=int, default=2, choices=(1, 2),
                    help="two E-steps in flight (trlda_model_set_stream_lanes, the bench's default): the "
                         "long document's tail runs under the next step's documents; 1: one launch at a time")
    args = ap.parse_args()
    import torch
    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments, DeviceBatch
    from trlda_amd.utils.synthetic import make_corpus
    L = _ffi.lib()
    _ffi.require_gpu()
    K, V, B = args.topics, args.words, args.batch
    dev = torch.device("cuda", 0)
    L.trlda_seed(1)
    lam = np.empty((K, V), order="F")
    L.trlda_sample_gamma_init(K, V, lam)
    model = _ffi.vp()
    _ffi.check(L.trlda_model_create(C.byref(model), 0, K, V))
    _ffi.check(L.trlda_model_set_stream(model, _ffi.vp(torch.cuda.current_stream(dev).cuda_stream)))
    _ffi.check(L.trlda_model_set_lambda(model, lam))
    _ffi.check(L.trlda_model_set_alpha(model, np.full(K, .1)))
    if args.wide:
        _ffi.check(L.trlda_model_set_doc_kernel(model, 2))
    _ffi.check(L.trlda_model_set_deferred_stats(model, int(not args.no_deferred)))
    g0 = np.empty((K, B), order="F")
    L.trlda_sample_gamma_init(K, B, g0)
    gamma0 = torch.from_numpy(np.ascontiguousarray(g0.T)).to(dev)
    _ffi.check(L.trlda_model_set_stream_lanes(model, 1 if args.no_deferred else args.lanes))
    outs = [(torch.empty(B * K, dtype=torch.float64, device=dev), torch.empty(K * V, dtype=torch.float64, device=dev))
            for _ in range(2)]
    up = (C.c_void_p * 2)()
    print("K=%d V=%d B=%d, %d steps per point, %d lane(s); us per E-step (documents kernel name)"
          % (K, V, B, args.steps, 1 if args.no_deferred else args.lanes))
    for series in args.series.split(","):
        for n in [int(x) for x in args.lengths.split(",")]:
            if n > V:
                continue
            lens = np.full(B, n) if series == "all" else np.concatenate([[n], np.full(B - 1, 100)])
            batches = [DeviceBatch(CSRDocuments(*make_corpus(B, V, seed=100 * n + i, lengths=lens)), V, 0)
                       for i in range(4)]

            def step(i):
                up[0] = batches[(i + 1) % 4].handle.value
                up[1] = batches[(i + 2) % 4].handle.value
                gamma, sstats = outs[i & 1]
                _ffi.check(L.trlda_model_estep_io_ahead(
                    model, batches[i % 4].handle, up, 2, gamma0.data_ptr(),
                    gamma.data_ptr(), sstats.data_ptr(), 20, 0., None))
            for i in range(10):
                step(i)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(args.steps):
                step(i)
            flags = L.trlda_model_last_deferred(model)
            _ffi.check(L.trlda_model_flush(model))    # (the last step's statistics: inside the clock)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t) / args.steps * 1e6
            print("%-4s n=%4d  %8.1f us  %s  fused_preamble=%d deferred=%d" % (
                series, n, us, L.trlda_model_last_doc_kernel(model).decode(),
                L.trlda_model_last_preamble_fused(model), flags), flush=True)
            for b in batches:
                b.close()


if __name__ == "__main__":
    main()
