"""Randomised check of DEFERRED statistics (csrc/estep_merged.h; GPU box; test infrastructure): a
random stream of trlda_model_estep_io_next calls over random batches -- shapes inside and outside
the stage's range (odd K, more than 256 documents, repeated ids with lists of more than 256
entries, documents of every tier incl. split ones, empty documents, zero counts), right, wrong and
missing announcements -- with, between the calls, everything that must flush what is pending
(trlda_model_flush, synchronize, lambda replaced, an update call, a batch closed and re-created,
the switch turned off and on, the model closed).  Every call's gamma, iteration counts and
statistics (from the array THAT call was given) must equal, bit for bit, those of the same stream
with the switch off; one call in three is also compared with the oracle.

`--lanes`: the stream goes through trlda_model_estep_io_ahead with two stream lanes
(trlda_model_set_stream_lanes: consecutive calls in flight at once on two streams of the library's
own), the batch after the next announced rightly, wrongly or not at all, the lanes switched off and
on in between -- against the same plain stream.

    python tests/fuzz_deferred.py [--cases 20] [--seed 1] [--lanes]   (tests/test_gpu_fuzz.py runs short ones)
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def random_batch(rng, V, B):
    kind = rng.randint(4)
    lens = []
    for _ in range(B):
        if kind == 0:
            n = rng.randint(0, 140)
        elif kind == 1:
            n = rng.choice([0, 1, 100, 128, 129, 144, 145, 192, 193, 300])
        elif kind == 2:
            n = int(np.exp(np.log(80) + .5 * rng.randn()))
        else:
            n = rng.randint(20, 60)
        lens.append(int(min(n, V if kind != 3 else 4 * V)))
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = np.concatenate([rng.permutation(V)[:n] if n <= V else rng.randint(0, V, size=n)
                          for n in lens] + [np.zeros(0, int)]).astype(np.int32)
    cnts = rng.randint(0, 4, size=ip[-1]).astype(np.int32)
    return ip, ids, cnts


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--calls", type=int, default=14)
    ap.add_argument("--lanes", action="store_true")
    args = ap.parse_args(argv)
    from oracle.pyoracle import Oracle                 # the checker
    import trlda_amd
    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.models import OnlineLDA
    L = _ffi.lib()
    orc = Oracle()
    rng = np.random.RandomState(args.seed)
    worst = 0.0
    deferred_calls = carried_calls = lane_calls = 0
    for case in range(args.cases):
        K = int(rng.choice([6, 7, 32, 100, 100, 128, 200]))
        V = int(rng.choice([40, 300, 2500, 7000]))
        n_b = 4
        Bs = [int(rng.choice([1, 17, 90, 200, 256, 300])) for _ in range(n_b)]
        raw = [random_batch(rng, V, B) for B in Bs]
        lams = [np.asfortranarray(rng.gamma(100., .01, (K, V))) for _ in range(2)]
        g0s = [np.asfortranarray(rng.gamma(100., .01, (K, B))) for B in Bs]
        max_iter = int(rng.choice([1, 5, 20]))
        # the script of the case: (op, args) -- the same for both runs
        script = []
        seq = [int(rng.randint(n_b)) for _ in range(args.calls + 2)]
        for c in range(args.calls):
            i = seq[c]
            r = rng.rand()                           # the announcement: right (60 %), wrong, none
            nxt = seq[c + 1] if r < .6 else int(rng.randint(n_b)) if r < .8 else -1
            nxt2 = -1
            if args.lanes:                           # ... and of the call after the next
                r = rng.rand()
                nxt2 = seq[c + 2] if r < .7 else int(rng.randint(n_b)) if r < .85 else -1
                if nxt < 0:
                    nxt, nxt2 = nxt2, -1             # (a list: no gap in front)
            script.append(("estep", i, int(nxt), int(nxt2)))
            if args.lanes and rng.rand() < .08:
                script.append(("lanes",))
            r = rng.rand()
            if r < .08:
                script.append(("flush",))
            elif r < .16:
                script.append(("sync",))
            elif r < .22:
                script.append(("lambda", int(rng.randint(2))))
            elif r < .28:
                script.append(("update", int(rng.randint(n_b)), int(rng.choice([0, 2]))))
            elif r < .34:
                script.append(("recreate", int(rng.randint(n_b))))
            elif r < .40:
                script.append(("toggle",))

        def run(deferred):
            nonlocal deferred_calls, carried_calls, lane_calls
            trlda_amd.seed(1000 + case)
            m = OnlineLDA(num_words=V, num_topics=K, num_documents=50000, alpha=.1, eta=.3)
            m.lambdas = lams[0]
            lam_now = [np.array(lams[0])]
            _ffi.check(L.trlda_model_set_deferred_stats(m._handle, deferred))
            dev = [m.upload(CSRDocuments(*r_)) for r_ in raw]
            outs, slots, on = [], [], bool(deferred)
            two = bool(deferred and args.lanes)
            if two:
                _ffi.check(L.trlda_model_set_stream_lanes(m._handle, 2))
            for step in script:
                if step[0] == "estep":
                    _, i, nxt, nxt2 = step
                    B = Bs[i]
                    ptrs = [_ffi.vp() for _ in range(4)]
                    for p, nbytes in zip(ptrs, (K * B * 8, K * B * 8, K * V * 8, B * 4)):
                        _ffi.check(L.trlda_dev_alloc(0, max(nbytes, 8), C.byref(p)))
                    _ffi.check(L.trlda_dev_upload(0, ptrs[0], g0s[i].ctypes.data, g0s[i].nbytes))
                    nan = np.full(K * V, np.nan)
                    _ffi.check(L.trlda_dev_upload(0, ptrs[2], nan.ctypes.data, nan.nbytes))
                    if deferred and args.lanes:
                        up = (C.c_void_p * 2)()
                        n_up = 0
                        for x in (nxt, nxt2):
                            if x >= 0:
                                up[n_up] = dev[x].handle.value
                                n_up += 1
                        _ffi.check(L.trlda_model_estep_io_ahead(m._handle, dev[i].handle, up, n_up, ptrs[0],
                                                                ptrs[1], ptrs[2], max_iter, 1e-3, ptrs[3]))
                    else:
                        _ffi.check(L.trlda_model_estep_io_next(m._handle, dev[i].handle,
                                                               dev[nxt].handle if nxt >= 0 else None,
                                                               ptrs[0], ptrs[1], ptrs[2], max_iter, 1e-3, ptrs[3]))
                    f = L.trlda_model_last_deferred(m._handle)
                    if deferred:
                        deferred_calls += f & 1
                        carried_calls += (f >> 1) & 1
                    slots.append((ptrs, i, np.array(lam_now[0])))
                elif step[0] == "flush":
                    _ffi.check(L.trlda_model_flush(m._handle))
                elif step[0] == "sync":
                    _ffi.check(L.trlda_model_synchronize(m._handle))
                elif step[0] == "lambda":
                    m.lambdas = lams[step[1]]
                    lam_now[0] = np.array(lams[step[1]])
                elif step[0] == "update":
                    m.update_parameters(dev[step[1]], max_iter_tr=step[2], max_iter_inference=max_iter)
                    lam_now[0] = np.array(m.lambdas)
                elif step[0] == "recreate":
                    dev[step[1]].close()
                    dev[step[1]] = m.upload(CSRDocuments(*raw[step[1]]))
                elif step[0] == "lanes":
                    if deferred:
                        two = not two
                        _ffi.check(L.trlda_model_set_stream_lanes(m._handle, 2 if two else 1))
                elif step[0] == "toggle":
                    on = not on
                    _ffi.check(L.trlda_model_set_deferred_stats(m._handle, int(on and deferred)))
            if deferred:
                lane_calls += L.trlda_model_lane_steps(m._handle)
            _ffi.check(L.trlda_model_synchronize(m._handle))
            for ptrs, i, lam_at in slots:
                B = Bs[i]
                g = np.empty((K, B), order="F"); s = np.empty((K, V), order="F"); it = np.empty(B, dtype=np.int32)
                _ffi.check(L.trlda_dev_download(0, g.ctypes.data, ptrs[1], g.nbytes))
                _ffi.check(L.trlda_dev_download(0, s.ctypes.data, ptrs[2], s.nbytes))
                _ffi.check(L.trlda_dev_download(0, it.ctypes.data, ptrs[3], it.nbytes))
                outs.append((g, s, it, i, lam_at))
                for p in ptrs:
                    L.trlda_dev_free(0, p)
            for d in dev:
                d.close()
            m.close()
            return outs

        a, b = run(1), run(0)
        assert len(a) == len(b)
        for n, (x, y) in enumerate(zip(a, b)):
            for q in range(3):
                if not np.array_equal(x[q], y[q]):
                    bad = np.argwhere(np.asarray(x[q]) != np.asarray(y[q]))
                    raise AssertionError("case %d call %d output %d differs from the plain stream: K %d V %d Bs %s "
                                         "max_iter %d batch %d; %d elements, first %s (%r against %r); script %s"
                                         % (case, n, q, K, V, Bs, max_iter, x[3], len(bad), bad[:3].tolist(),
                                            np.asarray(x[q])[tuple(bad[0])], np.asarray(y[q])[tuple(bad[0])], script))
            assert not np.isnan(x[1]).any(), "case %d call %d: statistics never written" % (case, n)
            if n % 3 == 0 and Bs[x[3]] > 0:
                ip, ids, cnts = raw[x[3]]
                go, so, ito = orc.estep(x[4], .1, ip, ids, cnts, g0s[x[3]], max_iter, 1e-3, nthreads=8)
                assert np.array_equal(x[2], ito), "case %d call %d: iteration counts" % (case, n)
                eg = float(np.max(np.abs(x[0] - go) / np.abs(go))) if go.size else 0.0
                big = so > 1e-150
                es = float(np.max(np.abs(x[1][big] - so[big]) / so[big])) if big.any() else 0.0
                worst = max(worst, eg, es)
                assert eg < 1e-8 and es < 1e-7, (case, n, eg, es)
        for r_, keep in zip(raw, [tuple(np.array(v) for v in r2) for r2 in raw]):
            assert all(np.array_equal(u, v) for u, v in zip(r_, keep))
    print("%d cases, %d calls left their statistics pending, %d launches carried them%s: all equal to the plain "
          "stream; worst against the oracle %.1e"
          % (args.cases, deferred_calls, carried_calls,
             ", %d calls went through the two lanes" % lane_calls if args.lanes else "", worst))
    if args.lanes:
        assert lane_calls > 0
    return worst


if __name__ == "__main__":
    main()
