#!/usr/bin/env python
"""Probe: the headline's stream of E-steps fed to ONE model on one stream, against the same
batches dealt alternately to P models (the same lambda) on P streams -- does the tail of one launch
(documents end 1-2 us apart; a 129..144-word document 5 us after the others) hide under the head of
the next when launches of two queues may overlap?  Prints us per step for P = 1, 2, 3."""
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
    K, V, B, NB = 100, 7000, 200, 200
    steps = int(os.environ.get("STEPS", "600"))
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
    for P in (1, 2, 3, 1, 2):
        models, streams, outs = [], [], []
        for p in range(P):
            m = _ffi.vp()
            _ffi.check(L.trlda_model_create(C.byref(m), 0, K, V))
            s = torch.cuda.Stream(device)
            _ffi.check(L.trlda_model_set_stream(m, _ffi.vp(s.cuda_stream)))
            _ffi.check(L.trlda_model_set_lambda(m, lam))
            _ffi.check(L.trlda_model_set_alpha(m, np.full(K, .1)))
            _ffi.check(L.trlda_model_set_deferred_stats(m, 1))
            models.append(m)
            streams.append(s)
            outs.append((torch.empty(B * K, dtype=torch.float64, device=device),
                         torch.empty(K * V, dtype=torch.float64, device=device)))

        nb = NB - NB % P                             # (every batch stays with one model)

        def run(first, n):
            for i in range(first, first + n):
                p = i % P
                j, nxt = i % nb, (i + P) % nb
                _ffi.check(L.trlda_model_estep_io_next(models[p], batches[j].handle, batches[nxt].handle,
                                                       g0s[j].data_ptr(), outs[p][0].data_ptr(),
                                                       outs[p][1].data_ptr(), 20, 0.0, None))

        def fence():
            for m in models:
                _ffi.check(L.trlda_model_flush(m))
            torch.cuda.synchronize()

        run(0, 3 * P)
        fence()
        pos = 3 * P
        for _ in range(4):                           # settle
            run(pos, 200)
            pos += 200
            fence()
        samples = []
        for _ in range(5):
            fence()
            t0 = time.perf_counter()
            run(pos, steps)
            fence()
            samples.append((time.perf_counter() - t0) / steps * 1e6)
            pos += steps
        print("P = %d streams: %s us per step (median %.2f)" % (P, " ".join("%.2f" % s for s in samples),
                                                                 sorted(samples)[2]), flush=True)
        for m in models:
            L.trlda_model_destroy(m)

    # the same through ONE model with two stream lanes (trlda_model_set_stream_lanes)
    for lanes, shared in ((1, False), (2, False), (2, True)):
        m = _ffi.vp()
        _ffi.check(L.trlda_model_create(C.byref(m), 0, K, V))
        s = torch.cuda.current_stream(device)
        _ffi.check(L.trlda_model_set_stream(m, _ffi.vp(s.cuda_stream)))
        _ffi.check(L.trlda_model_set_lambda(m, lam))
        _ffi.check(L.trlda_model_set_alpha(m, np.full(K, .1)))
        _ffi.check(L.trlda_model_set_deferred_stats(m, 1))
        _ffi.check(L.trlda_model_set_stream_lanes(m, lanes))
        outs = [(torch.empty(B * K, dtype=torch.float64, device=device),
                 torch.empty(K * V, dtype=torch.float64, device=device)) for _ in range(2)]
        up = (C.c_void_p * 2)()

        def run(first, n):
            for i in range(first, first + n):
                j = i % NB
                up[0] = batches[(i + 1) % NB].handle.value
                up[1] = batches[(i + 2) % NB].handle.value
                o = outs[0 if shared else i & 1]
                _ffi.check(L.trlda_model_estep_io_ahead(m, batches[j].handle, up, 2, g0s[j].data_ptr(),
                                                        o[0].data_ptr(), o[1].data_ptr(), 20, 0.0, None))

        def fence():
            _ffi.check(L.trlda_model_flush(m))
            torch.cuda.synchronize()

        run(0, 6)
        fence()
        pos = 6
        for _ in range(4):
            run(pos, 200)
            pos += 200
            fence()
        samples = []
        for _ in range(5):
            fence()
            t0 = time.perf_counter()
            run(pos, steps)
            fence()
            samples.append((time.perf_counter() - t0) / steps * 1e6)
            pos += steps
        if os.environ.get("TRACE_CALLS"):            # where a slow region loses its time: call by call
            fence()
            ts = []
            for i in range(pos, pos + steps):
                t0 = time.perf_counter()
                run(i, 1)
                ts.append((time.perf_counter() - t0) * 1e6)
            t0 = time.perf_counter(); fence(); tf = (time.perf_counter() - t0) * 1e6
            pos += steps
            ts = np.array(ts)
            slow = np.nonzero(ts > 150)[0]
            print("   calls: median %.1f us, sum %.0f us, fence %.0f us; calls > 150 us: %s" % (
                np.median(ts), ts.sum(), tf, [(int(i), int(ts[i])) for i in slow[:20]]))
        two, one = C.c_double(), C.c_double()
        L.trlda_model_lane_timing(m, C.byref(two), C.byref(one))
        print("one model, %d lane(s)%s: %s us per step (median %.2f); %d steps through the lanes; lane state %d, "
              "the library's own measurement: a launch %.2f us, a step %.2f us"
              % (lanes, ", ONE set of output arrays" if shared else "", " ".join("%.2f" % x for x in samples),
                 sorted(samples)[2], L.trlda_model_lane_steps(m), L.trlda_model_lane_state(m), two.value, one.value),
              flush=True)
        L.trlda_model_destroy(m)


if __name__ == "__main__":
    main()
