#!/usr/bin/env python
"""Probe: P models on P streams, the headline's batches dealt in turn, in the three forms of a step --
deferred statistics + announced preamble (one launch per step), announced preamble only (documents
launch + statistics kernel), neither (preamble kernel + documents launch + statistics kernel): do the
small kernels of one stream hide under the documents of the other?"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments, DeviceBatch
    from trlda_amd.utils.synthetic import SEED_BASE, make_corpus
    L = _ffi.lib()
    _ffi.require_gpu()
    K, V, B, NB = [int(x) for x in os.environ.get("SHAPE", "100,7000,200,120").split(",")]
    steps = int(os.environ.get("STEPS", "600"))
    max_p = int(os.environ.get("MAXP", "4"))
    device = torch.device("cuda", 0)
    L.trlda_seed(1)
    lam = np.empty((K, V), order="F")
    L.trlda_sample_gamma_init(K, V, lam)
    csrs = [CSRDocuments(*make_corpus(B, V, seed=SEED_BASE + 1 + i, mean_unique=100)) for i in range(NB)]
    batches = [DeviceBatch(c, V, 0) for c in csrs]
    g0s = []
    for i in range(NB):
        g0 = np.empty((K, B), order="F")
        L.trlda_sample_gamma_init(K, B, g0)
        g0s.append(torch.from_numpy(np.ascontiguousarray(g0.T)).to(device))
    for deferred, announce in ((1, 1), (0, 1), (0, 0)):
        for P in range(1, max_p + 1):
            models, streams, outs = [], [], []
            for p in range(P):
                m = _ffi.vp()
                _ffi.check(L.trlda_model_create(C.byref(m), 0, K, V))
                s = torch.cuda.Stream(device, priority=-1)
                _ffi.check(L.trlda_model_set_stream(m, _ffi.vp(s.cuda_stream)))
                _ffi.check(L.trlda_model_set_lambda(m, lam))
                _ffi.check(L.trlda_model_set_alpha(m, np.full(K, .1)))
                _ffi.check(L.trlda_model_set_deferred_stats(m, deferred))
                models.append(m)
                streams.append(s)
                outs.append((torch.empty(B * K, dtype=torch.float64, device=device),
                             torch.empty(K * V, dtype=torch.float64, device=device)))
            nb = NB - NB % P

            def run(first, n):
                for i in range(first, first + n):
                    p = i % P
                    j, nxt = i % nb, (i + P) % nb
                    _ffi.check(L.trlda_model_estep_io_next(models[p], batches[j].handle,
                                                           batches[nxt].handle if announce else None,
                                                           g0s[j].data_ptr(), outs[p][0].data_ptr(),
                                                           outs[p][1].data_ptr(), 20, 1e-3, None))

            def fence():
                for m in models:
                    _ffi.check(L.trlda_model_flush(m))
                torch.cuda.synchronize()

            run(0, 3 * P)
            fence()
            pos = 3 * P
            for _ in range(3):
                run(pos, min(200, steps))
                pos += min(200, steps)
                fence()
            samples = []
            for _ in range(3):
                fence()
                t0 = time.perf_counter()
                run(pos, steps)
                fence()
                samples.append((time.perf_counter() - t0) / steps * 1e6)
                pos += steps
            print("deferred %d announced %d  P = %d streams: %s us per step"
                  % (deferred, announce, P, " ".join("%.2f" % s for s in samples)), flush=True)
            for m in models:
                L.trlda_model_destroy(m)


if __name__ == "__main__":
    main()
