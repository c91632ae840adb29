"""Stream lanes (trlda_model_set_stream_lanes, trlda_model_estep_io_ahead): the E-steps of a deferred
stream -- a corpus pass of LDA::updateVariablesVI calls on an unchanged lambda, reference
src/lda.cpp:160-220, whose calls share nothing but lambda -- dealt in turn to two streams of the
library's own, so that consecutive calls' launches overlap on the device.

Checked here, through the C ABI: a stream through two lanes against the same stream one launch at a
time and against the plain E-step -- bitwise the same gamma, statistics and iteration counts, every
call's results in the arrays THAT call was given; against the oracle; what joins the lanes (flush,
synchronize, lambda replaced, an update, a batch destroyed, the model destroyed); callers that hand
consecutive calls the same arrays or chain gamma into the next gamma0 (one lane, still right); work
the caller enqueued on the model's stream just before a call; batches a lane does not take; wrong
and missing announcements."""
import ctypes as C

import numpy as np
import pytest

from helpers import TIGHT_RTOL, HipSampler, relerr, seeded_gamma, seeded_lambda
from test_gpu_deferred import Slots, corpus, make_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(hip_lib):
    from trlda_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a visible MI355X"
    return hip_lib


@pytest.fixture(scope="module")
def sampler(hip):
    return HipSampler(hip)


def ahead(hip, m, devs, slots, n, announce=2, max_iter=20):
    """call n of the stream `devs`, the next `announce` batches announced"""
    from trlda_amd import _ffi
    up = (C.c_void_p * 2)()
    k = 0
    for a in range(announce):
        if n + 1 + a < len(devs):
            up[k] = devs[n + 1 + a].handle.value
            k += 1
    g0d, gd, sd, itd = slots[n].ptrs
    _ffi.check(hip.trlda_model_estep_io_ahead(m._handle, devs[n].handle, up, k, g0d, gd, sd, max_iter,
                                              1e-3, itd))


def run_stream(hip, K, V, lam, csrs, g0s, order, lanes, deferred=1, announce=2, max_iter=20):
    from trlda_amd import _ffi
    m = make_model(K, V, lam)
    dev = [m.upload(c) for c in csrs]
    slots = [Slots(hip, K, V, csrs[i], g0s[i]) for i in order]
    devs = [dev[i] for i in order]
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, deferred))
    _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, lanes))
    for n in range(len(order)):
        ahead(hip, m, devs, slots, n, announce, max_iter)
    through = hip.trlda_model_lane_steps(m._handle)
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    res = [s.read() for s in slots]
    for s in slots:
        s.free()
    m.close()
    return res, through


@pytest.mark.parametrize("K,V,B", [(100, 7000, 200), (64, 900, 90), (128, 3000, 256)])
def test_a_stream_through_two_lanes_equals_the_one_lane_stream(hip, oracle, sampler, K, V, B):
    lam = seeded_lambda(sampler, 3, K, V)
    lens = [None, None, None, np.r_[[129, 140, 150], np.full(B - 3, 60)], None,
            np.r_[[200, 400], np.full(B - 2, 80)]]      # tiered launches, a split document
    csrs = [corpus(B - (i % 2) * 7, V, seed=40 + i, mean_unique=min(100, V // 8),
                   lengths=None if l is None else l[:B - (i % 2) * 7]) for i, l in enumerate(lens)]
    g0s = [seeded_gamma(sampler, 50 + i, K, len(c)) for i, c in enumerate(csrs)]
    order = [0, 1, 2, 3, 4, 5, 0, 3, 1, 2, 2, 5]
    two, through = run_stream(hip, K, V, lam, csrs, g0s, order, lanes=2)
    assert through == len(order)
    one, through1 = run_stream(hip, K, V, lam, csrs, g0s, order, lanes=1)
    assert through1 == 0
    plain, _ = run_stream(hip, K, V, lam, csrs, g0s, order, lanes=1, deferred=0, announce=0)
    for n in range(len(order)):
        for q in range(3):
            assert np.array_equal(two[n][q], one[n][q]), (n, q)
            assert np.array_equal(two[n][q], plain[n][q]), (n, q)
        assert not np.isnan(two[n][1]).any()
    for n in (0, 3, 5, 11):
        c, g0 = csrs[order[n]], g0s[order[n]]
        go, so, ito = oracle.estep(lam, .1, c.indptr, c.ids, c.cnts, g0, 20, 1e-3, nthreads=8)
        g, s, it = two[n]
        assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL
        assert relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL and np.array_equal(s == 0, so == 0)


@pytest.mark.parametrize("announce", [0, 1, 2])
def test_announcements_missing_short_or_wrong(hip, sampler, announce):
    """a lane prepares the preamble of ITS next call from upcoming[1]: without it (or with a promise
    that is not kept) the call fills its own -- the same results"""
    from trlda_amd import _ffi
    K, V, B = 100, 3000, 120
    lam = seeded_lambda(sampler, 5, K, V)
    csrs = [corpus(B, V, seed=70 + i, mean_unique=60) for i in range(5)]
    g0s = [seeded_gamma(sampler, 80 + i, K, B) for i in range(5)]
    order = [0, 1, 2, 3, 4, 2, 0]
    ref, _ = run_stream(hip, K, V, lam, csrs, g0s, order, lanes=1, deferred=0, announce=0)
    got, through = run_stream(hip, K, V, lam, csrs, g0s, order, lanes=2, announce=announce)
    assert through == len(order)
    for n in range(len(order)):
        for q in range(3):
            assert np.array_equal(got[n][q], ref[n][q]), (n, q)
    # promises that are not kept: every call announces batches 3 and 4, whatever comes
    m = make_model(K, V, lam)
    dev = [m.upload(c) for c in csrs]
    slots = [Slots(hip, K, V, csrs[i], g0s[i]) for i in order]
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, 2))
    up = (C.c_void_p * 2)(dev[3].handle.value, dev[4].handle.value)
    for n, i in enumerate(order):
        g0d, gd, sd, itd = slots[n].ptrs
        _ffi.check(hip.trlda_model_estep_io_ahead(m._handle, dev[i].handle, up, 2, g0d, gd, sd, 20, 1e-3, itd))
    _ffi.check(hip.trlda_model_flush(m._handle))
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    for n in range(len(order)):
        r = slots[n].read()
        for q in range(3):
            assert np.array_equal(r[q], ref[n][q]), (n, q)
        slots[n].free()
    m.close()


def test_what_joins_the_lanes(hip, oracle, sampler):
    """between calls of a stream through the lanes: flush, synchronize, a host read of the statistics'
    array after a flush, lambda replaced (the lanes' prepared preambles are void), an update call, the
    batch destroyed while a lane's statistics still read it -- and the model destroyed with both lanes
    busy: every call's results complete and right"""
    from trlda_amd import _ffi
    K, V, B = 100, 4000, 150
    lam = seeded_lambda(sampler, 7, K, V)
    lam2 = seeded_lambda(sampler, 8, K, V)
    csrs = [corpus(B, V, seed=90 + i, mean_unique=80) for i in range(4)]
    g0s = [seeded_gamma(sampler, 95 + i, K, B) for i in range(4)]

    def want(l, i):
        return oracle.estep(l, .1, csrs[i].indptr, csrs[i].ids, csrs[i].cnts, g0s[i], 20, 1e-3, nthreads=8)

    def check(slot, l, i):
        g, s, it = slot.read()
        go, so, ito = want(l, i)
        assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL
        assert relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL and np.array_equal(s == 0, so == 0)

    m = make_model(K, V, lam)
    dev = [m.upload(c) for c in csrs]
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, 2))
    order = [0, 1, 2, 3, 0, 1, 2, 3, 0, 1]
    slots = [Slots(hip, K, V, csrs[i], g0s[i]) for i in order]
    devs = [dev[i] for i in order]
    ahead(hip, m, devs, slots, 0)
    ahead(hip, m, devs, slots, 1)
    _ffi.check(hip.trlda_model_flush(m._handle))               # flush: both complete on the model's stream
    _ffi.check(hip.trlda_dev_synchronize(0))
    check(slots[0], lam, 0)
    check(slots[1], lam, 1)
    ahead(hip, m, devs, slots, 2)
    ahead(hip, m, devs, slots, 3)
    ahead(hip, m, devs, slots, 4)
    m.lambdas = lam2                                           # joins; the prepared preambles are void
    check(slots[2], lam, 2)
    check(slots[3], lam, 3)
    check(slots[4], lam, 0)
    ahead(hip, m, devs, slots, 5)
    ahead(hip, m, devs, slots, 6)
    through = hip.trlda_model_lane_steps(m._handle)
    assert through == 7
    m.update_parameters(dev[3], max_iter_tr=2, max_iter_inference=20)      # an update joins, lambda moves
    lam3 = np.asfortranarray(m.lambdas)
    # (the update's kernels left lambda's row sums behind and E-steps on that lambda use them -- one
    # lane; lambda set anew: two again)
    ahead(hip, m, devs, slots, 7)
    assert hip.trlda_model_lane_steps(m._handle) == through
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    check(slots[7], lam3, 3)
    slots[7].poison()
    m.lambdas = lam3
    check(slots[5], lam2, 1)
    check(slots[6], lam2, 2)
    ahead(hip, m, devs, slots, 7)
    ahead(hip, m, devs, slots, 8)
    # the batch of call 8 goes while a lane's statistics still read it
    csr_x = corpus(B, V, seed=333, mean_unique=80)
    g0x = seeded_gamma(sampler, 334, K, B)
    bx = m.upload(csr_x)
    sx = Slots(hip, K, V, csr_x, g0x)
    up = (C.c_void_p * 2)()
    _ffi.check(hip.trlda_model_estep_io_ahead(m._handle, bx.handle, up, 0, sx.ptrs[0], sx.ptrs[1], sx.ptrs[2],
                                              20, 1e-3, sx.ptrs[3]))
    bx.close()
    ahead(hip, m, devs, slots, 9)
    # (call 7 again into the arrays of the call before it: one lane, by the rule for shared arrays)
    assert hip.trlda_model_lane_steps(m._handle) == through + 3
    m.close()                                                  # both lanes busy
    _ffi.check(hip.trlda_dev_synchronize(0))
    check(slots[7], lam3, 3)
    check(slots[8], lam3, 0)
    check(slots[9], lam3, 1)
    g, s, it = sx.read()
    go, so, ito = oracle.estep(lam3, .1, csr_x.indptr, csr_x.ids, csr_x.cnts, g0x, 20, 1e-3, nthreads=8)
    assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL and relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL
    for s_ in slots + [sx]:
        s_.free()


def test_shared_and_chained_arrays_take_one_lane(hip, sampler):
    """the same output arrays for consecutive calls, or the last call's gamma as this call's gamma0:
    nothing to run side by side -- such calls go one at a time, with the results of the plain stream"""
    from trlda_amd import _ffi
    K, V, B = 100, 2500, 100
    lam = seeded_lambda(sampler, 11, K, V)
    csrs = [corpus(B, V, seed=120 + i, mean_unique=70) for i in range(3)]
    g0 = seeded_gamma(sampler, 130, K, B)
    n_calls = 7

    def stream(lanes, deferred, mode):
        m = make_model(K, V, lam)
        dev = [m.upload(c) for c in csrs]
        _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, deferred))
        _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, lanes))
        slots = [Slots(hip, K, V, csrs[0], g0) for _ in range(n_calls + 1)]
        up = (C.c_void_p * 2)()
        out = []
        for n in range(n_calls):
            up[0] = dev[(n + 1) % 3].handle.value
            up[1] = dev[(n + 2) % 3].handle.value
            if mode == "shared":                     # one set of arrays for every call
                g0d, gd, sd, itd = slots[0].ptrs
            else:                                    # chained: gamma of call n - 1 is gamma0 of call n
                g0d = slots[n].ptrs[1] if n else slots[0].ptrs[0]
                gd, sd, itd = slots[n + 1].ptrs[1:]
            _ffi.check(hip.trlda_model_estep_io_ahead(m._handle, dev[n % 3].handle, up, 2, g0d, gd, sd, 20,
                                                      1e-3, itd))
            if mode == "shared":
                _ffi.check(hip.trlda_model_flush(m._handle))
                _ffi.check(hip.trlda_dev_synchronize(0))
                out.append(slots[0].read())
        through = hip.trlda_model_lane_steps(m._handle)
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        if mode == "chained":
            out = [slots[n + 1].read() for n in range(n_calls)]
        for s in slots:
            s.free()
        m.close()
        return out, through

    for mode in ("chained", "shared"):
        ref, _ = stream(1, 0, mode)
        got, through = stream(2, 1, mode)
        for n in range(n_calls):
            for q in range(3):
                assert np.array_equal(got[n][q], ref[n][q]), (mode, n, q)
        if mode == "chained":
            assert through <= 1, through             # (the first call has no predecessor)

    # ... and WITHOUT a flush in between, shared arrays: the last call's results are what is left
    m = make_model(K, V, lam)
    dev = [m.upload(c) for c in csrs]
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, 2))
    s = Slots(hip, K, V, csrs[0], g0)
    up = (C.c_void_p * 2)()
    for n in range(6):
        up[0] = dev[(n + 1) % 3].handle.value
        up[1] = dev[(n + 2) % 3].handle.value
        _ffi.check(hip.trlda_model_estep_io_ahead(m._handle, dev[n % 3].handle, up, 2, s.ptrs[0], s.ptrs[1],
                                                  s.ptrs[2], 20, 1e-3, s.ptrs[3]))
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    last = s.read()
    s.poison()
    _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, 1))
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 0))
    _ffi.check(hip.trlda_model_estep_io(m._handle, dev[5 % 3].handle, s.ptrs[0], s.ptrs[1], s.ptrs[2], 20, 1e-3,
                                        s.ptrs[3]))
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    want = s.read()
    for q in range(3):
        assert np.array_equal(last[q], want[q]), q
    s.free()
    m.close()


@pytest.mark.parametrize("n_sets", [3, 5, 7])
def test_an_odd_rotation_of_output_sets_meets_the_other_lane(hip, sampler, n_sets):
    """with deferred statistics the sstats array of call i is written by call i + 2's
    launch on the same lane.  A caller that rotates an ODD number of output sets hands the set of
    call i to call i + n_sets, which goes to the OTHER lane -- the library keeps every range a lane
    has been given since the last join, so that call waits for the lane that may still write the set.
    Every call's results are read (after a flush) before its set comes round again; the last
    round's are read at the end: all equal to the plain stream's."""
    from trlda_amd import _ffi
    K, V, B = 100, 2500, 100
    lam = seeded_lambda(sampler, 12, K, V)
    csrs = [corpus(B, V, seed=140 + i, mean_unique=70) for i in range(4)]
    g0 = seeded_gamma(sampler, 150, K, B)
    n_calls = 4 * n_sets + 1

    def stream(lanes, deferred, check_every):
        m = make_model(K, V, lam)
        dev = [m.upload(c) for c in csrs]
        _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, deferred))
        _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, lanes))
        slots = [Slots(hip, K, V, csrs[0], g0) for _ in range(n_sets)]
        up = (C.c_void_p * 2)()
        out = [None] * n_calls
        for n in range(n_calls):
            up[0] = dev[(n + 1) % 4].handle.value
            up[1] = dev[(n + 2) % 4].handle.value
            g0d, gd, sd, itd = slots[n % n_sets].ptrs
            _ffi.check(hip.trlda_model_estep_io_ahead(m._handle, dev[n % 4].handle, up, 2, g0d, gd, sd, 20,
                                                      1e-3, itd))
            if check_every and n % n_sets == n_sets - 1:     # a whole round is out: read it
                _ffi.check(hip.trlda_model_flush(m._handle))
                _ffi.check(hip.trlda_model_synchronize(m._handle))
                for q in range(n - n_sets + 1, n + 1):
                    out[q] = slots[q % n_sets].read()
        through = hip.trlda_model_lane_steps(m._handle)
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        for q in range(n_calls):
            if out[q] is None and q + n_sets >= n_calls:     # the sets as the last calls left them
                out[q] = slots[q % n_sets].read()
        for s in slots:
            s.free()
        m.close()
        return out, through

    ref, _ = stream(1, 0, True)
    for check_every in (True, False):
        got, through = stream(2, 1, check_every)
        assert through == n_calls
        for n in range(n_calls):
            if got[n] is None:
                continue
            for q in range(3):
                assert np.array_equal(got[n][q], ref[n][q]), (n_sets, check_every, n, q)


def test_the_callers_stream_comes_first(hip, sampler):
    """gamma0 produced by work on the model's stream right before the call (a device copy enqueued on
    torch's current stream, which the model is told to use), results consumed on that stream after
    trlda_model_flush without a host synchronisation in between"""
    import torch
    from trlda_amd import _ffi
    K, V, B = 100, 3000, 128
    lam = seeded_lambda(sampler, 13, K, V)
    csrs = [corpus(B, V, seed=140 + i, mean_unique=90) for i in range(4)]
    g0s = [seeded_gamma(sampler, 150 + i, K, B) for i in range(4)]
    dev_t = torch.device("cuda", 0)
    m = make_model(K, V, lam)
    stream = torch.cuda.Stream(dev_t)
    _ffi.check(hip.trlda_model_set_stream(m._handle, _ffi.vp(stream.cuda_stream)))
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, 2))
    devb = [m.upload(c) for c in csrs]
    host_g0 = [torch.from_numpy(np.ascontiguousarray(g.T)).pin_memory() for g in g0s]
    big = torch.empty(64 << 20, dtype=torch.float64, device=dev_t)      # keeps the stream busy
    g0_dev = [torch.empty(B * K, dtype=torch.float64, device=dev_t) for _ in range(4)]
    gam = [torch.empty(B * K, dtype=torch.float64, device=dev_t) for _ in range(4)]
    sst = [torch.full((K * V,), float("nan"), dtype=torch.float64, device=dev_t) for _ in range(4)]
    sums = []
    up = (C.c_void_p * 2)()
    with torch.cuda.stream(stream):
        for n in range(4):
            big.fill_(float(n))                                       # ~ a millisecond of work in front
            g0_dev[n].copy_(host_g0[n].reshape(-1), non_blocking=True)
            k = 0
            for a in (1, 2):
                if n + a < 4:
                    up[k] = devb[n + a].handle.value
                    k += 1
            _ffi.check(hip.trlda_model_estep_io_ahead(m._handle, devb[n].handle, up, k, g0_dev[n].data_ptr(),
                                                      gam[n].data_ptr(), sst[n].data_ptr(), 20, 1e-3, None))
        assert hip.trlda_model_lane_steps(m._handle) == 4
        _ffi.check(hip.trlda_model_flush(m._handle))
        for n in range(4):
            sums.append((gam[n].sum(), sst[n].sum()))                 # on the stream, behind the join
    stream.synchronize()
    want, _ = run_stream(hip, K, V, lam, csrs, g0s, [0, 1, 2, 3], lanes=1, deferred=0, announce=0)
    for n in range(4):
        gg = np.asfortranarray(gam[n].cpu().numpy().reshape(B, K).T)
        ss = np.asfortranarray(sst[n].cpu().numpy().reshape(V, K).T)
        assert np.array_equal(gg, want[n][0]) and np.array_equal(ss, want[n][1]), n
        # what the caller's own kernels saw on the stream behind the join: the finished arrays
        assert float(sums[n][0]) == float(gam[n].sum()) and float(sums[n][1]) == float(sst[n].sum())
        assert np.isfinite(float(sums[n][1]))
    m.close()


@pytest.mark.parametrize("K,V,B,why", [(101, 600, 50, "odd K"), (100, 3000, 300, "more than 256 documents"),
                                       (200, 2000, 64, "K > 128"), (100, 7000, 1600, "config 3's batch")])
def test_shapes_outside_the_deferred_range_go_through_the_lanes_too(hip, sampler, K, V, B, why):
    """their steps are several kernels (preamble, documents, statistics) on the lane's stream, with
    the deferred statistics switched on or off -- the results of the plain stream either way"""
    lam = seeded_lambda(sampler, 17, K, V)
    csrs = [corpus(B, V, seed=170 + i, mean_unique=min(80, V // 8)) for i in range(3)]
    g0s = [seeded_gamma(sampler, 175 + i, K, B) for i in range(3)]
    order = [0, 1, 2, 0, 1]
    ref, _ = run_stream(hip, K, V, lam, csrs, g0s, order, lanes=1, deferred=0, announce=0)
    for deferred in (1, 0):
        got, through = run_stream(hip, K, V, lam, csrs, g0s, order, lanes=2, deferred=deferred)
        assert through == len(order), why
        for n in range(len(order)):
            for q in range(3):
                assert np.array_equal(got[n][q], ref[n][q]), (why, deferred, n, q)


def test_atomic_statistics_and_a_dense_preamble_through_the_lanes(hip, sampler):
    from trlda_amd import _ffi
    K, V, B = 100, 2000, 120
    lam = seeded_lambda(sampler, 23, K, V)
    csrs = [corpus(B, V, seed=270 + i, mean_unique=60) for i in range(3)]
    g0s = [seeded_gamma(sampler, 275 + i, K, B) for i in range(3)]
    order = [0, 1, 2, 1]

    def stream(lanes, mode, dense):
        m = make_model(K, V, lam)
        dev = [m.upload(c) for c in csrs]
        slots = [Slots(hip, K, V, csrs[i], g0s[i]) for i in order]
        devs = [dev[i] for i in order]
        _ffi.check(hip.trlda_model_set_sstats_mode(m._handle, mode))
        _ffi.check(hip.trlda_model_set_dense_preamble(m._handle, dense))
        _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
        _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, lanes))
        for n in range(len(order)):
            ahead(hip, m, devs, slots, n)
        through = hip.trlda_model_lane_steps(m._handle)
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        res = [s.read() for s in slots]
        for s in slots:
            s.free()
        m.close()
        return res, through

    for mode, dense in ((1, 0), (0, 1)):
        ref, _ = stream(1, mode, dense)
        got, through = stream(2, mode, dense)
        assert through == len(order)
        for n in range(len(order)):
            assert np.array_equal(got[n][0], ref[n][0]) and np.array_equal(got[n][2], ref[n][2])
            if mode == 1:                            # atomic additions: the order is not fixed
                assert relerr(got[n][1], ref[n][1]) < TIGHT_RTOL
            else:
                assert np.array_equal(got[n][1], ref[n][1])


def test_four_hundred_steps_through_two_lanes(hip, sampler):
    """a long stream over eight batches, two sets of output arrays in turn (bench.py's step): the
    last two calls' results against the plain E-step, and the counters' wrap-free bookkeeping"""
    from trlda_amd import _ffi
    K, V, B = 100, 7000, 200
    lam = seeded_lambda(sampler, 19, K, V)
    csrs = [corpus(B, V, seed=200 + i) for i in range(8)]
    g0s = [seeded_gamma(sampler, 210 + i, K, B) for i in range(8)]
    m = make_model(K, V, lam)
    dev = [m.upload(c) for c in csrs]
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, 2))
    g0_slots = [Slots(hip, K, V, csrs[i], g0s[i]) for i in range(8)]
    outs = [Slots(hip, K, V, csrs[0], g0s[0]) for _ in range(2)]
    up = (C.c_void_p * 2)()
    N = 401
    for n in range(N):
        up[0] = dev[(n + 1) % 8].handle.value
        up[1] = dev[(n + 2) % 8].handle.value
        o = outs[n & 1]
        _ffi.check(hip.trlda_model_estep_io_ahead(m._handle, dev[n % 8].handle, up, 2, g0_slots[n % 8].ptrs[0],
                                                  o.ptrs[1], o.ptrs[2], 20, 1e-3, o.ptrs[3]))
        if n % 97 == 50:
            _ffi.check(hip.trlda_model_flush(m._handle))
    # (after 96 steps through the lanes the library times one launch of a lane and the four behind it,
    # and keeps the lanes if a launch lasts well over a step -- include/trlda_hip.h, trlda_model_lane_state)
    through = hip.trlda_model_lane_steps(m._handle)
    state = hip.trlda_model_lane_state(m._handle)
    assert state in (1, 2) and (through == N if state == 2 else 96 <= through < N), (state, through)
    launch, stepus = C.c_double(), C.c_double()
    _ffi.check(hip.trlda_model_lane_timing(m._handle, C.byref(launch), C.byref(stepus)))
    if launch.value > 0:                             # (0: every window was host-bound -- no verdict, lanes kept)
        assert 5. < launch.value < 1000. and 5. < stepus.value < 500., (launch.value, stepus.value)
        # (the lanes go when two of the last four looks said no -- launches that do not overlap, or two
        # lanes not faster than one: one look that said no leaves them, so the last measurement alone
        # does not tell the state; lanes that are gone have had a look that said no)
    else:
        assert state == 2
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    got = {(N - 1) & 1: outs[(N - 1) & 1].read(), (N - 2) & 1: outs[(N - 2) & 1].read()}
    _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, 1))
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 0))
    for n in (N - 1, N - 2):
        s = Slots(hip, K, V, csrs[n % 8], g0s[n % 8])
        _ffi.check(hip.trlda_model_estep_io(m._handle, dev[n % 8].handle, s.ptrs[0], s.ptrs[1], s.ptrs[2], 20,
                                            1e-3, s.ptrs[3]))
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        want = s.read()
        for q in range(3):
            assert np.array_equal(got[n & 1][q], want[q]), (n, q)
        s.free()
    for s in g0_slots + outs:
        s.free()
    m.close()


def test_the_python_stream_equals_a_loop_of_do_e_step(hip, sampler):
    """trlda_amd.stream.EStepStream on torch tensors against the reference's own form of the pass:
    a Python loop over do_e_step with host arrays (python/src/ldainterface.cpp:311-390)"""
    import torch
    from trlda_amd.stream import EStepStream
    K, V, B = 100, 3000, 160
    lam = seeded_lambda(sampler, 29, K, V)
    csrs = [corpus(B, V, seed=300 + i, mean_unique=80) for i in range(5)]
    g0s = [seeded_gamma(sampler, 310 + i, K, B) for i in range(5)]
    m = make_model(K, V, lam)
    dev = torch.device("cuda", 0)
    batches = [m.upload(c) for c in csrs]
    g0_t = [torch.from_numpy(np.ascontiguousarray(g.T)).to(dev) for g in g0s]
    gam = [torch.empty(B, K, dtype=torch.float64, device=dev) for _ in csrs]
    sst = [torch.full((V, K), float("nan"), dtype=torch.float64, device=dev) for _ in csrs]
    its = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in csrs]
    with EStepStream(m) as s:
        for i, b in enumerate(batches):
            s.step(b, batches[i + 1:i + 3], g0_t[i], gam[i], sst[i], max_iter=20, iterations=its[i])
        through = s.steps_through_lanes
    torch.cuda.synchronize()
    assert through == len(csrs)
    for i, c in enumerate(csrs):
        g, ss, it = m.do_e_step(c, latents=g0s[i], max_iter=20, return_iterations=True)
        assert np.array_equal(gam[i].cpu().numpy().T, np.asarray(g))
        assert np.array_equal(sst[i].cpu().numpy().T, np.asarray(ss))
        assert np.array_equal(its[i].cpu().numpy(), np.asarray(it))
    m.close()


def test_lanes_that_do_not_pay_are_given_up(hip, sampler, monkeypatch):
    """the lanes' gain rests on the runtime's hardware queues.  The library looks at
    the streams it makes (lanes_ensure) and MEASURES: after 96 steps through the lanes one launch of a
    lane is timed together with the four behind it; two launches in flight means a launch lasts about
    two steps, and lanes whose launches last less than one step are dropped -- the stream goes on
    one launch at a time (trlda_model_lane_state 1).  Here the bar is put where the lanes must lose
    (TRLDA_LANE_CAL_MIN_IN_FLIGHT=100) and where they must win (0): the state, the step counter that
    stops, and every call's results bitwise those of the plain stream -- through the lanes and after
    they are gone."""
    from trlda_amd import _ffi
    K, V, B = 100, 3000, 120
    lam = seeded_lambda(sampler, 23, K, V)
    csrs = [corpus(B, V, seed=300 + i, mean_unique=60) for i in range(5)]
    g0s = [seeded_gamma(sampler, 310 + i, K, B) for i in range(5)]
    N, S = 400, 20                                   # (a look: a stretch that ends early, then a window; two looks say no)
    ref, _ = run_stream(hip, K, V, lam, csrs, g0s, [q % 5 for q in range(S)], lanes=1, deferred=0, announce=0)
    monkeypatch.setenv("TRLDA_LANE_CAL_HOST_SHARE", "1e9")   # (this loop reads arrays back between calls: its
    for bar, want_state in (("100", 1), ("0", 2)):           #  windows would otherwise count as host-bound)
        monkeypatch.setenv("TRLDA_LANE_CAL_MIN_IN_FLIGHT", bar)
        m = make_model(K, V, lam)
        dev = [m.upload(c) for c in csrs]
        _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
        _ffi.check(hip.trlda_model_set_stream_lanes(m._handle, 2))
        slots = [Slots(hip, K, V, csrs[q % 5], g0s[q % 5]) for q in range(S)]
        up = (C.c_void_p * 2)()
        for n in range(N):
            up[0] = dev[(n + 1) % 5].handle.value
            up[1] = dev[(n + 2) % 5].handle.value
            g0d, gd, sd, itd = slots[n % S].ptrs     # (S is a multiple of 5: a set always meets the same batch)
            _ffi.check(hip.trlda_model_estep_io_ahead(m._handle, dev[n % 5].handle, up, 2, g0d, gd, sd, 20, 1e-3, itd))
            if n % 40 == 39:                         # the last twenty calls' results
                _ffi.check(hip.trlda_model_flush(m._handle))
                _ffi.check(hip.trlda_model_synchronize(m._handle))
                for q in range(S):
                    r = slots[q].read()
                    for w in range(3):
                        assert np.array_equal(r[w], ref[q][w]), (bar, n, q, w)
        through = hip.trlda_model_lane_steps(m._handle)
        state = hip.trlda_model_lane_state(m._handle)
        launch, stepus = C.c_double(), C.c_double()
        _ffi.check(hip.trlda_model_lane_timing(m._handle, C.byref(launch), C.byref(stepus)))
        assert state == want_state, (bar, state, through, launch.value, stepus.value)
        assert launch.value > 0 and stepus.value > 0
        assert (96 <= through < N - 40) if want_state == 1 else through == N, (bar, through)
        for s_ in slots:
            s_.free()
        m.close()
