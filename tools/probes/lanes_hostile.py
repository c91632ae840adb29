#!/usr/bin/env python
"""Probe: the two stream lanes in a process that is hostile to them -- a dozen
streams made first (torch's, default and high priority, all of them used and alive), the model on
a torch side stream -- against one lane and against the same batches dealt by hand to two models.
The library looks at the streams it makes (a probe kernel on each of two: side by side or one behind
the other) and keeps a pair that overlaps; where none is to be had it goes one launch at a time.

    python tools/probes/lanes_hostile.py [--hostile 12] [--lengths lognormal]

Prints us per step (median of 5 regions of STEPS steps) for: one lane | two lanes (library) | two
models by hand, and trlda_model_lane_state."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hostile", type=int, default=12)
    ap.add_argument("--lengths", default="poisson")
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    import torch
    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments, DeviceBatch
    from trlda_amd.utils.synthetic import SEED_BASE, make_corpus
    L = _ffi.lib()
    _ffi.require_gpu()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # ---- the hostile part: streams of both priorities, used, alive for the whole run
    others = []
    for i in range(a.hostile):
        s = torch.cuda.Stream(device, priority=-1 if i % 2 else 0)
        with torch.cuda.stream(s):
            torch.zeros(64, device=device).add_(1.)
        others.append(s)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device)                 # the model lives on a torch side stream
    K, V, B, NB = 100, 7000, 200, 200
    L.trlda_seed(1)
    lam = np.empty((K, V), order="F")
    L.trlda_sample_gamma_init(K, V, lam)
    rng = np.random.RandomState(5)
    csrs = []
    for i in range(NB):
        if a.lengths == "lognormal":
            lens = np.clip(np.round(rng.lognormal(np.log(90.), .6, B)), 5, 1500).astype(int)
            csrs.append(CSRDocuments(*make_corpus(B, V, seed=SEED_BASE + 1 + i, mean_unique=100, lengths=lens)))
        else:
            csrs.append(CSRDocuments(*make_corpus(B, V, seed=SEED_BASE + 1 + i, mean_unique=100)))
    batches = [DeviceBatch(c, V, 0) for c in csrs]
    g0s = []
    for i in range(NB):
        g0 = np.empty((K, B), order="F")
        L.trlda_sample_gamma_init(K, B, g0)
        g0s.append(torch.from_numpy(np.ascontiguousarray(g0.T)).to(device))
    outs = [(torch.empty(B * K, dtype=torch.float64, device=device),
             torch.empty(K * V, dtype=torch.float64, device=device)) for _ in range(2)]
    up = (C.c_void_p * 2)()

    def model_on(stream):
        m = _ffi.vp()
        _ffi.check(L.trlda_model_create(C.byref(m), 0, K, V))
        _ffi.check(L.trlda_model_set_stream(m, _ffi.vp(stream.cuda_stream)))
        _ffi.check(L.trlda_model_set_lambda(m, lam))
        _ffi.check(L.trlda_model_set_alpha(m, np.full(K, .1)))
        _ffi.check(L.trlda_model_set_deferred_stats(m, 1))
        return m

    def timed(run, fence, steps):
        pos = 0
        for _ in range(4):
            run(pos, steps); pos += steps; fence()
        ts = []
        for _ in range(5):
            fence(); t0 = time.perf_counter()
            run(pos, steps); fence()
            ts.append((time.perf_counter() - t0) / steps * 1e6); pos += steps
        return ts

    res = {}
    for lanes in (1, 2):
        m = model_on(side)
        _ffi.check(L.trlda_model_set_stream_lanes(m, lanes))

        def run(first, n, m=m):
            for i in range(first, first + n):
                j = i % NB
                up[0] = batches[(i + 1) % NB].handle.value
                up[1] = batches[(i + 2) % NB].handle.value
                o = outs[i & 1]
                _ffi.check(L.trlda_model_estep_io_ahead(m, batches[j].handle, up, 2, g0s[j].data_ptr(),
                                                        o[0].data_ptr(), o[1].data_ptr(), 20, 1e-3, None))

        def fence(m=m):
            _ffi.check(L.trlda_model_flush(m)); torch.cuda.synchronize()
        ts = timed(run, fence, a.steps)
        res[lanes] = ts
        print("one model, %d lane(s): %s us per step (median %.2f); lane state %d, %d steps through the lanes" % (
            lanes, " ".join("%.2f" % t for t in ts), float(np.median(ts)), L.trlda_model_lane_state(m),
            L.trlda_model_lane_steps(m)), flush=True)
        L.trlda_model_destroy(m)
    # ---- the same batches dealt by hand to two models on two fresh streams
    ms = [model_on(torch.cuda.Stream(device)) for _ in range(2)]
    nb = NB - NB % 2

    def run2(first, n):
        for i in range(first, first + n):
            p = i % 2
            j, nxt = i % nb, (i + 2) % nb
            _ffi.check(L.trlda_model_estep_io_next(ms[p], batches[j].handle, batches[nxt].handle,
                                                   g0s[j].data_ptr(), outs[p][0].data_ptr(),
                                                   outs[p][1].data_ptr(), 20, 1e-3, None))

    def fence2():
        for m in ms:
            _ffi.check(L.trlda_model_flush(m))
        torch.cuda.synchronize()
    ts = timed(run2, fence2, a.steps)
    print("two models by hand (two fresh torch streams): %s us per step (median %.2f)" % (
        " ".join("%.2f" % t for t in ts), float(np.median(ts))))
    print("two lanes / by hand: %.3f;  two lanes / one lane: %.3f" % (
        np.median(res[2]) / np.median(ts), np.median(res[2]) / np.median(res[1])))
    del others


if __name__ == "__main__":
    main()
