"""Deferred statistics (csrc/estep_merged.h, trlda_model_set_deferred_stats): in a stream of
E-steps on an unchanged lambda -- a corpus pass of LDA::updateVariablesVI calls, reference
src/lda.cpp:160-220 -- the statistics of a call (:207-217) are formed by extra workgroups of the
NEXT call's document launch, or by the kernel of their own as soon as anything else touches the
model.

Checked here, through the C ABI: a stream with the switch on against the same stream with it off
-- bitwise the same gamma, statistics and iteration counts, every call's statistics in the array
THAT call was given; against the oracle; which calls deferred and which launches carried
(trlda_model_last_deferred); everything that must flush (trlda_model_flush, synchronize, lambda
replaced, an update call, the batch destroyed, the model destroyed); batches outside the stage's
range; wrong and missing announcements."""
import ctypes as C

import numpy as np
import pytest

from helpers import TIGHT_RTOL, HipSampler, relerr, seeded_gamma, seeded_lambda

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(hip_lib):
    from trlda_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a visible MI355X"
    return hip_lib


@pytest.fixture(scope="module")
def sampler(hip):
    return HipSampler(hip)


def make_model(K, V, lam, D=100000):
    from trlda_amd.models import OnlineLDA
    m = OnlineLDA(num_words=V, num_topics=K, num_documents=D, alpha=.1, eta=.3)
    m.lambdas = lam
    return m


def corpus(B, V, seed, mean_unique=100, lengths=None):
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    return CSRDocuments(*make_corpus(B, V, seed=seed, mean_unique=mean_unique, lengths=lengths))


class Slots(object):
    """device arrays (gamma0, gamma, sstats, iterations) for one batch; sstats starts as NaN"""

    def __init__(self, hip, K, V, csr, g0):
        from trlda_amd import _ffi
        self.hip, self.K, self.V, self.B = hip, K, V, len(csr)
        n = K * self.B
        self.ptrs = [_ffi.vp() for _ in range(4)]
        for p, nbytes in zip(self.ptrs, (n * 8, n * 8, K * V * 8, max(self.B, 1) * 4)):
            _ffi.check(hip.trlda_dev_alloc(0, max(nbytes, 8), C.byref(p)))
        _ffi.check(hip.trlda_dev_upload(0, self.ptrs[0], g0.ctypes.data, n * 8))
        self.poison()

    def poison(self):
        from trlda_amd import _ffi
        nan = np.full(self.K * self.V, np.nan)
        _ffi.check(self.hip.trlda_dev_upload(0, self.ptrs[2], nan.ctypes.data, nan.nbytes))

    def read(self):
        from trlda_amd import _ffi
        g = np.empty((self.K, self.B), order="F")
        s = np.empty((self.K, self.V), order="F")
        it = np.empty(self.B, dtype=np.int32)
        _ffi.check(self.hip.trlda_dev_download(0, g.ctypes.data, self.ptrs[1], g.nbytes))
        _ffi.check(self.hip.trlda_dev_download(0, s.ctypes.data, self.ptrs[2], s.nbytes))
        _ffi.check(self.hip.trlda_dev_download(0, it.ctypes.data, self.ptrs[3], it.nbytes))
        return g, s, it

    def free(self):
        for p in self.ptrs:
            self.hip.trlda_dev_free(0, p)


def estep(hip, m, dev, slots, i, nxt, max_iter=20):
    from trlda_amd import _ffi
    g0d, gd, sd, itd = slots[i].ptrs
    _ffi.check(hip.trlda_model_estep_io_next(m._handle, dev[i].handle,
                                             dev[nxt].handle if nxt is not None else None,
                                             g0d, gd, sd, max_iter, 1e-3, itd))
    return hip.trlda_model_last_deferred(m._handle)


@pytest.mark.parametrize("K,V,B", [(100, 7000, 200), (64, 900, 90), (128, 3000, 256)])
def test_a_deferred_stream_equals_the_plain_one(hip, oracle, sampler, K, V, B):
    from trlda_amd import _ffi
    lam = seeded_lambda(sampler, 3, K, V)
    lens = [None, None, None, np.r_[[129, 140, 150], np.full(B - 3, 60)], None,
            np.r_[[200, 400], np.full(B - 2, 80)]]      # tiered launches, a split document
    csrs = [corpus(B - (i % 2) * 7, V, seed=40 + i, mean_unique=min(100, V // 8),
                   lengths=None if l is None else l[:B - (i % 2) * 7]) for i, l in enumerate(lens)]
    g0s = [seeded_gamma(sampler, 50 + i, K, len(c)) for i, c in enumerate(csrs)]
    order = [0, 1, 2, 3, 4, 5, 0, 3, 1]
    results = {}
    for deferred in (1, 0):
        m = make_model(K, V, lam)
        dev = [m.upload(c) for c in csrs]
        # one set of output arrays PER CALL: a call's statistics must land in the array it was given
        slots = [Slots(hip, K, V, csrs[i], g0s[i]) for i in order]
        devs = [dev[i] for i in order]
        _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, deferred))
        flags = []
        for n in range(len(order)):
            flags.append(estep(hip, m, devs, slots, n, n + 1 if n + 1 < len(order) else None))
        _ffi.check(hip.trlda_model_synchronize(m._handle))     # flushes the last one
        results[deferred] = [s.read() for s in slots]
        if deferred:
            # every call of the stream left its statistics pending; every launch but the first
            # carried the call before's (the first call's preamble is a launch of its own into the
            # model's buffer, which the second call -- prefetched -- does not touch)
            assert all(f & 1 for f in flags), flags
            assert all(f & 2 for f in flags[1:]), flags
            assert hip.trlda_model_last_deferred(m._handle) & 1
        else:
            assert flags == [0] * len(order)
        for s in slots:
            s.free()
        m.close()
    for n, (a, b) in enumerate(zip(results[1], results[0])):
        for q in range(3):
            assert np.array_equal(a[q], b[q]), (n, q)
        assert not np.isnan(a[1]).any()
    for n in (0, 3, 5):
        c, g0 = csrs[order[n]], g0s[order[n]]
        go, so, ito = oracle.estep(lam, .1, c.indptr, c.ids, c.cnts, g0, 20, 1e-3, nthreads=8)
        g, s, it = results[1][n]
        assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL
        assert relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL and np.array_equal(s == 0, so == 0)


@pytest.mark.parametrize("K,V,B,n", [(100, 12, 60, 9), (10, 40, 256, 3), (128, 500, 200, 1)])
def test_deferred_stage_corner_shapes(hip, oracle, sampler, K, V, B, n):
    """Every list long (V = 12: the stage has no short-list workgroups, the long-list ones write the
    zero columns), lists of up to 256 entries, one word per document."""
    from trlda_amd import _ffi
    lam = seeded_lambda(sampler, 3, K, V)
    csrs = [corpus(B, V, seed=60 + i, lengths=np.full(B, n)) for i in range(3)]
    g0s = [seeded_gamma(sampler, 70 + i, K, B) for i in range(3)]
    m = make_model(K, V, lam)
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    dev = [m.upload(c) for c in csrs]
    slots = [Slots(hip, K, V, c, g) for c, g in zip(csrs, g0s)]
    flags = [estep(hip, m, dev, slots, i, (i + 1) % 3) for i in range(3)]
    assert flags == [1, 3, 3], flags
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    for i in range(3):
        c = csrs[i]
        go, so, ito = oracle.estep(lam, .1, c.indptr, c.ids, c.cnts, g0s[i], 20, 1e-3, nthreads=8)
        g, s, it = slots[i].read()
        assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL
        assert relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL and np.array_equal(s == 0, so == 0)
        slots[i].free()
    m.close()


def test_what_must_flush_does(hip, oracle, sampler):
    """The statistics of a deferred call are in their array after trlda_model_flush + a wait for
    the stream -- and after anything else that touches the model or the batch."""
    import trlda_amd
    from trlda_amd import _ffi
    K, V, B = 100, 3000, 120
    lam = seeded_lambda(sampler, 5, K, V)
    lam2 = seeded_lambda(sampler, 6, K, V)
    csrs = [corpus(B, V, seed=70 + i, mean_unique=60) for i in range(3)]
    g0s = [seeded_gamma(sampler, 80 + i, K, B) for i in range(3)]
    want = [oracle.estep(lam, .1, c.indptr, c.ids, c.cnts, g, 20, 1e-3, nthreads=8) for c, g in zip(csrs, g0s)]

    def check(slot, n):
        g, s, it = slot.read()
        go, so, ito = want[n]
        assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL
        assert relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL and np.array_equal(s == 0, so == 0)

    m = make_model(K, V, lam)
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    dev = [m.upload(c) for c in csrs]
    slots = [Slots(hip, K, V, c, g) for c, g in zip(csrs, g0s)]

    # trlda_model_flush: enqueues, does not wait
    assert estep(hip, m, dev, slots, 0, 1) & 1
    _ffi.check(hip.trlda_model_flush(m._handle))
    _ffi.check(hip.trlda_dev_synchronize(0))          # (the device, not the model: nothing flushes again)
    check(slots[0], 0)
    # an unannounced batch next: the pending statistics are launched first, the new call defers
    slots[0].poison()
    assert estep(hip, m, dev, slots, 0, None) & 1
    f = estep(hip, m, dev, slots, 2, None)
    assert f & 1 and not f & 2                       # (not carried: this call fills the model's own buffer)
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    check(slots[0], 0)
    check(slots[2], 2)
    # the batch destroyed while its statistics are pending
    slots[1].poison()
    tmp = m.upload(csrs[1])
    g0d, gd, sd, itd = slots[1].ptrs
    _ffi.check(hip.trlda_model_estep_io_next(m._handle, tmp.handle, None, g0d, gd, sd, 20, 1e-3, itd))
    assert hip.trlda_model_last_deferred(m._handle) & 1
    tmp.close()
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    check(slots[1], 1)
    # lambda replaced behind a pending call: the statistics are those of the OLD lambda
    slots[0].poison()
    assert estep(hip, m, dev, slots, 0, 1) & 1
    m.lambdas = lam2
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    check(slots[0], 0)
    m.lambdas = lam
    # an update call behind a pending call (and a deferred call behind the update: lambda moved,
    # nothing stale is used)
    slots[2].poison()
    assert estep(hip, m, dev, slots, 2, 0) & 1
    trlda_amd.seed(3)
    m.update_parameters(dev[1], max_iter_tr=2, max_iter_inference=20)
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    check(slots[2], 2)
    lam_after = np.array(m.lambdas)
    slots[0].poison()
    assert estep(hip, m, dev, slots, 0, None) & 1
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    go, so, ito = oracle.estep(lam_after, .1, csrs[0].indptr, csrs[0].ids, csrs[0].cnts, g0s[0], 20, 1e-3,
                               nthreads=8)
    g, s, it = slots[0].read()
    assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL and relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL
    # the switch turned off with statistics pending; the model destroyed with statistics pending
    m.lambdas = lam
    slots[1].poison()
    assert estep(hip, m, dev, slots, 1, None) & 1
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 0))
    assert estep(hip, m, dev, slots, 2, None) == 0
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    check(slots[1], 1)
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    slots[0].poison()
    assert estep(hip, m, dev, slots, 0, None) & 1
    m.close()                                        # (destroy: flushes, then waits)
    check(slots[0], 0)
    for s in slots:
        s.free()


def test_outside_the_stage_nothing_is_deferred(hip, oracle, sampler):
    from trlda_amd import _ffi
    V = 2000
    cases = [(100, 300, 40, "more than 256 documents"), (7, 50, 30, "odd K"), (200, 50, 30, "K > 128")]
    for K, B, mean, why in cases:
        lam = seeded_lambda(sampler, 9, K, V)
        m = make_model(K, V, lam)
        _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
        csrs = [corpus(B, V, seed=5 + i, mean_unique=mean) for i in range(2)]
        g0s = [seeded_gamma(sampler, 7 + i, K, B) for i in range(2)]
        dev = [m.upload(c) for c in csrs]
        slots = [Slots(hip, K, V, c, g) for c, g in zip(csrs, g0s)]
        assert estep(hip, m, dev, slots, 0, 1) == 0, why
        assert estep(hip, m, dev, slots, 1, 0) == 0, why
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        for n in range(2):
            go, so, ito = oracle.estep(lam, .1, csrs[n].indptr, csrs[n].ids, csrs[n].cnts, g0s[n], 20, 1e-3,
                                       nthreads=8)
            g, s, it = slots[n].read()
            assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL
            assert relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL
        for s in slots:
            s.free()
        m.close()
    # a stream that alternates a batch inside the range with one outside it: the one inside defers,
    # the (announced, prefetched) launch of the one outside carries them and launches its own
    # statistics as a kernel
    K = 100
    lam = seeded_lambda(sampler, 9, K, V)
    m = make_model(K, V, lam)
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    csrs = [corpus(100, V, seed=1, mean_unique=40), corpus(300, V, seed=2, mean_unique=40)]
    g0s = [seeded_gamma(sampler, 3 + i, K, len(c)) for i, c in enumerate(csrs)]
    dev = [m.upload(c) for c in csrs]
    slots = [Slots(hip, K, V, c, g) for c, g in zip(csrs, g0s)]
    flags = [estep(hip, m, dev, slots, i, 1 - i) for i in (0, 1, 0, 1)]
    assert flags == [1, 2, 1, 2], flags
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    for n in range(2):
        go, so, ito = oracle.estep(lam, .1, csrs[n].indptr, csrs[n].ids, csrs[n].cnts, g0s[n], 20, 1e-3,
                                   nthreads=8)
        g, s, it = slots[n].read()
        assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL and relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL
    for s in slots:
        s.free()
    m.close()


def test_many_deferred_steps_in_a_row(hip, sampler):
    """400 steps over four batches and three rotating exp(psi(lambda)) buffers without a
    synchronisation in between: every batch's statistics equal its first ones, bitwise."""
    from trlda_amd import _ffi
    K, V, B = 100, 7000, 200
    lam = seeded_lambda(sampler, 17, K, V)
    m = make_model(K, V, lam)
    csrs = [corpus(B - 3 * i, V, seed=20 + i) for i in range(4)]
    g0s = [seeded_gamma(sampler, 30 + i, K, len(c)) for i, c in enumerate(csrs)]
    dev = [m.upload(c) for c in csrs]
    slots = [Slots(hip, K, V, c, g) for c, g in zip(csrs, g0s)]
    first = []
    for i in range(4):
        estep(hip, m, dev, slots, i, None)
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        first.append(slots[i].read())
        slots[i].poison()
    _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    for n in range(400):
        f = estep(hip, m, dev, slots, n % 4, (n + 1) % 4)
        assert f & 1 and (n == 0 or f & 2)
    _ffi.check(hip.trlda_model_synchronize(m._handle))
    for i in range(4):
        got = slots[i].read()
        for q in range(3):
            assert np.array_equal(got[q], first[i][q]), (i, q)
    for s in slots:
        s.free()
    m.close()


def test_two_models_with_pending_statistics_on_one_batch(hip, oracle, sampler):
    """A batch remembers one model whose pending statistics read it; when a second model defers on
    the same batch the first one's statistics are launched, and closing the batch launches the
    second's -- nobody reads a recycled allocation."""
    from trlda_amd import _ffi
    K, V, B = 100, 3000, 150
    lams = [seeded_lambda(sampler, 41 + i, K, V) for i in range(2)]
    csr = corpus(B, V, seed=9, mean_unique=60)
    g0 = seeded_gamma(sampler, 43, K, B)
    ms = [make_model(K, V, lam) for lam in lams]
    for m in ms:
        _ffi.check(hip.trlda_model_set_deferred_stats(m._handle, 1))
    shared = ms[0].upload(csr)
    slots = [Slots(hip, K, V, csr, g0) for _ in ms]
    for m, s in zip(ms, slots):
        g0d, gd, sd, itd = s.ptrs
        _ffi.check(hip.trlda_model_estep_io_next(m._handle, shared.handle, None, g0d, gd, sd, 20, 1e-3, itd))
        assert hip.trlda_model_last_deferred(m._handle) & 1
    shared.close()                                   # (model 1's statistics are still pending here)
    other = ms[0].upload(corpus(B, V, seed=10, mean_unique=60))      # may recycle the allocation
    for m in ms:
        _ffi.check(hip.trlda_model_synchronize(m._handle))
    for lam, s in zip(lams, slots):
        go, so, ito = oracle.estep(lam, .1, csr.indptr, csr.ids, csr.cnts, g0, 20, 1e-3, nthreads=8)
        g, st, it = s.read()
        assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL
        assert relerr(st[so > 0], so[so > 0]) < TIGHT_RTOL and np.array_equal(st == 0, so == 0)
        s.free()
    other.close()
    for m in ms:
        m.close()
