"""Heavy-tailed and residency cases of the document launch, pinned as tests (round 4):

* the reference's own performance workload (code/trlda/python/tests/onlinelda_test.py:204-246,
  test_speed) with parity instead of a stopwatch;
* split documents when the launch has more segment workgroups than the chip has CUs and the
  segment counts (3, 5, 7) do not divide anything;
* a NaN in a split document's gamma0: NaN results at once, like the reference (lda.cpp:176-204),
  not a wait for rows that "never arrive".
"""
import time

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


def make_model(K, V, lam, alpha=.1, eta=.3, D=10000):
    from trlda_amd.models import OnlineLDA
    m = OnlineLDA(num_words=V, num_topics=K, num_documents=D, alpha=alpha, eta=eta)
    m.lambdas = lam
    return m


def check_sstats(got, want, rtol=TIGHT_RTOL):
    assert relerr(got[want > 0], want[want > 0]) < rtol
    assert np.array_equal(got == 0, want == 0)


def test_reference_speed_workload(hip, oracle, sampler):
    """onlinelda_test.py:204-246 (test_speed): K = 100, W = 1000, 110 documents of 1..600 unique
    words with counts 0..9 (zero counts are legal), gamma0 ~ Gamma(100, 1/100),
    do_e_step(max_iter=100) from a list of tuples -- the reference only asserts that it beats
    Hoffman's NumPy code; here gamma, the statistics and every document's iteration count against
    the oracle, with the long documents split over workgroups and without."""
    K, V, D, N = 100, 1000, 110, 600
    rng = np.random.RandomState(20150706)
    docs = []
    for _ in range(D):                              # onlinelda_test.py:228-233
        wordids = rng.permutation(V)[:1 + rng.randint(N)]
        docs.append([(int(w), int(rng.randint(10))) for w in wordids])
    g0 = np.asfortranarray(rng.gamma(100., 1. / 100., (K, D)))
    from trlda_amd.documents import as_csr
    c = as_csr(docs)
    lens = np.diff(c.indptr)
    assert lens.max() > 500 and lens.min() < 20    # the heavy tail is there
    lam = seeded_lambda(sampler, 1, K, V)
    m = make_model(K, V, lam)
    go, so, ito = oracle.estep(lam, .1, c.indptr, c.ids, c.cnts, g0, 100, 1e-3, nthreads=8)
    assert ito.min() < 100                          # documents leave through the convergence test
    for split in (1, 0):
        assert hip.trlda_model_set_split_docs(m._handle, split) == 0
        g, s, it = m.do_e_step(docs, max_iter=100, latents=g0, return_iterations=True)
        assert hip.trlda_model_last_doc_kernel(m._handle) == b"estep_docs_tiered_kernel"
        assert (hip.trlda_model_last_split_workgroups(m._handle) > 0) == bool(split)
        assert np.array_equal(it, ito)
        per_doc = np.max(np.abs(g - go) / np.abs(go), axis=0)
        assert per_doc.max() < TIGHT_RTOL, sorted(zip(per_doc, lens))[-3:]
        check_sstats(s, so)
        assert hip.trlda_model_synchronize(m._handle) == 0


@pytest.mark.parametrize("K", [100, 64])
def test_split_documents_beyond_the_resident_workgroups(hip, oracle, sampler, K):
    """More segment workgroups than CUs (300 on 256), with 3, 5 and 7 segments per document in an
    order that puts documents across the boundary of what is resident at once: the segments
    of a document wait for peers that are dispatched later (the dispatcher hands out workgroups
    in order and the workgroups it waits for belong to documents that finish).  Parity, equal
    iteration counts, no give-up (the synchronising call succeeds), bitwise run to run."""
    from trlda_amd.documents import CSRDocuments
    V = 3000
    rng = np.random.RandomState(11 + K)
    lens = []
    for _ in range(20):
        lens += [800, 600, 300]                     # 7, 5, 3 segments of <= 128 words
    lens += [40, 100, 128, 150]
    lens = np.array(lens)
    rng.shuffle(lens)
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = np.concatenate([rng.permutation(V)[:n] for n in lens]).astype(np.int32)
    cnts = (1 + rng.randint(3, size=ip[-1])).astype(np.int32)
    docs = CSRDocuments(ip, ids, cnts)
    lam = seeded_lambda(sampler, 91, K, V)
    g0 = seeded_gamma(sampler, 92, K, len(lens))
    m = make_model(K, V, lam)
    go, so, ito = oracle.estep(lam, .1, ip, ids, cnts, g0, 30, 1e-3, nthreads=8)
    want_wgs = int(sum(-(-n // 128) - 1 for n in lens if n > 192))
    assert want_wgs + len(lens) > 256 + 40
    once = None
    for rep in range(3):
        g, s, it = m.update_variables(docs, latents=g0, max_iter=30, threshold=1e-3,
                                      return_iterations=True)
        assert hip.trlda_model_last_split_workgroups(m._handle) == want_wgs
        assert hip.trlda_model_synchronize(m._handle) == 0      # xerr == 0: nobody gave up
        if once is None:
            once = (g, s)
            assert np.array_equal(it, ito)
            per_doc = np.max(np.abs(g - go) / np.abs(go), axis=0)
            assert per_doc.max() < TIGHT_RTOL, sorted(zip(per_doc, lens))[-3:]
            check_sstats(s, so)
        else:
            assert np.array_equal(g, once[0]) and np.array_equal(s, once[1])


def test_nan_in_a_split_document_is_a_result_not_a_wait(hip, sampler):
    """the exchange rows of a split document start out as the all-ones bit pattern; a
    partial sum that IS NaN (a caller's gamma0 holds one) used to look like a row that never
    arrived -- every segment then waited 2^21 polls per iteration.  Now it arrives as a NaN: the
    document's gamma is NaN at once (the reference returns NaNs too, lda.cpp:176-204), the other
    documents are untouched, the call succeeds and takes milliseconds."""
    from trlda_amd.documents import CSRDocuments
    K, V = 100, 2000
    rng = np.random.RandomState(5)
    lens = [600, 90, 400, 120]
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = np.concatenate([rng.permutation(V)[:n] for n in lens]).astype(np.int32)
    cnts = np.ones(ip[-1], np.int32)
    docs = CSRDocuments(ip, ids, cnts)
    lam = seeded_lambda(sampler, 17, K, V)
    g0 = seeded_gamma(sampler, 18, K, len(lens))
    m = make_model(K, V, lam)
    clean = m.update_variables(docs, latents=g0, max_iter=20, threshold=1e-3)
    assert hip.trlda_model_last_split_workgroups(m._handle) > 0
    bad = g0.copy(order="F")
    bad[3, 0] = np.nan
    # the all-ones pattern itself, as an input: a quiet NaN like any other
    bad[5, 2] = np.frombuffer(np.array([-1], np.int64).tobytes(), np.float64)[0]
    t0 = time.perf_counter()
    g, s = m.update_variables(docs, latents=bad, max_iter=20, threshold=1e-3)
    dt = time.perf_counter() - t0
    assert hip.trlda_model_synchronize(m._handle) == 0
    assert dt < 1.0, dt
    assert np.all(np.isnan(g[:, 0])) and np.all(np.isnan(g[:, 2]))
    assert np.array_equal(g[:, 1], clean[0][:, 1]) and np.array_equal(g[:, 3], clean[0][:, 3])
