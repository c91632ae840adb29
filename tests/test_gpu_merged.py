"""The merged launch (csrc/estep_merged.h): on small tables the statistics of an E-step
(reference src/lda.cpp:207-217), the M-step (src/onlinelda.cpp:99-100, src/batchlda.cpp:60) and
the row sums of the next E-step (src/lda.cpp:172) are extra workgroups of the DOCUMENT launch that
wait for a documents-done counter -- one launch per E-step, one per trust-region iteration.

Checked here: against the stand-alone statistics kernel (trlda_model_set_merged_launch(model, 0))
-- bitwise the same statistics, gamma and iteration counts -- against the oracle, run to run, over
hundreds of consecutive launches (the counters only grow), for every document-kernel variant a
merged launch can carry (registers, 144-word variant, single orientation, split documents), and
that whatever is outside its range still takes the kernel of its own."""
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


def make_model(K, V, lam, alpha=.1, eta=.3, D=10000):
    from trlda_amd.models import OnlineLDA
    m = OnlineLDA(num_words=V, num_topics=K, num_documents=D, alpha=alpha, eta=eta)
    m.lambdas = lam
    return m


def corpus(B, V, seed, mean_unique=60, lengths=None):
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    return CSRDocuments(*make_corpus(B, V, seed=seed, mean_unique=mean_unique, lengths=lengths))


def check_sstats(got, want, rtol=TIGHT_RTOL):
    assert relerr(got[want > 0], want[want > 0]) < rtol
    assert np.array_equal(got == 0, want == 0)


@pytest.mark.parametrize("K,V,B,mean", [(100, 7000, 200, 100), (128, 3000, 64, 50), (64, 500, 224, 30),
                                        (10, 1000, 100, 40), (2, 60, 7, 10), (100, 7000, 1, 100)])
def test_estep_merged_equals_the_kernel_of_its_own(hip, oracle, sampler, K, V, B, mean):
    docs = corpus(B, V, seed=100 + K + B, mean_unique=mean)
    lam = seeded_lambda(sampler, 3, K, V)
    g0 = seeded_gamma(sampler, 4, K, B)
    m = make_model(K, V, lam)
    res = {}
    for merged in (1, 0):
        assert hip.trlda_model_set_merged_launch(m._handle, 2 * merged) == 0     # 2: plain E-steps too
        res[merged] = m.update_variables(docs, latents=g0, max_iter=20, threshold=1e-3,
                                         return_iterations=True)
        assert hip.trlda_model_last_merged(m._handle) == merged
        assert hip.trlda_model_last_preamble_fused(m._handle) == 1
    (g1, s1, i1), (g0_, s0, i0) = res[1], res[0]
    assert np.array_equal(g1, g0_) and np.array_equal(i1, i0)
    assert np.array_equal(s1, s0)                    # same sums in the same order: bitwise
    go, so, ito = oracle.estep(lam, .1, docs.indptr, docs.ids, docs.cnts, g0, 20, 1e-3, nthreads=8)
    assert np.array_equal(i1, ito) and relerr(g1, go) < TIGHT_RTOL
    check_sstats(s1, so)
    # run to run, and the counters after many launches
    hip.trlda_model_set_merged_launch(m._handle, 2)
    for _ in range(40):
        g, s = m.update_variables(docs, latents=g0, max_iter=20, threshold=1e-3)
        assert hip.trlda_model_last_merged(m._handle) == 1
    assert np.array_equal(g, g1) and np.array_equal(s, s1)
    assert hip.trlda_model_synchronize(m._handle) == 0


def test_merged_launch_carries_every_document_variant(hip, oracle, sampler):
    """A merged launch whose documents take the register body, the 144-word variant, the single
    orientation (150..192 words, and 2100 words: LDS rows and streamed rows) and SEGMENTS of split
    documents: each of them stores for the statistics stage and counts itself done."""
    from trlda_amd.documents import CSRDocuments
    K, V = 100, 3000
    rng = np.random.RandomState(8)
    lens = [5, 0, 100, 128, 130, 144, 150, 192, 193, 300, 600, 1, 64, 2100]
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = np.concatenate([rng.permutation(V)[:n] for n in lens]).astype(np.int32)
    cnts = rng.randint(0, 4, size=ip[-1]).astype(np.int32)           # zero counts are legal
    docs = CSRDocuments(ip, ids, cnts)
    lam = seeded_lambda(sampler, 5, K, V)
    g0 = seeded_gamma(sampler, 6, K, len(lens))
    m = make_model(K, V, lam)
    go, so, ito = oracle.estep(lam, .1, ip, ids, cnts, g0, 30, 1e-3, nthreads=8)
    for split in (1, 0):
        hip.trlda_model_set_split_docs(m._handle, split)
        res = {}
        for merged in (1, 0):
            hip.trlda_model_set_merged_launch(m._handle, 2 * merged)
            res[merged] = m.update_variables(docs, latents=g0, max_iter=30, threshold=1e-3,
                                             return_iterations=True)
            assert hip.trlda_model_last_merged(m._handle) == merged
            assert hip.trlda_model_last_doc_kernel(m._handle) == b"estep_docs_tiered_kernel"
        assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1])
        assert np.array_equal(res[1][2], ito)
        per_doc = np.max(np.abs(res[1][0] - go) / np.abs(go), axis=0)
        assert per_doc.max() < TIGHT_RTOL, list(zip(lens, per_doc))
        check_sstats(res[1][1], so)
    assert hip.trlda_model_synchronize(m._handle) == 0


def test_outside_its_range_the_statistics_stay_a_kernel_of_their_own(hip, sampler):
    K, V = 100, 2000
    lam = seeded_lambda(sampler, 7, K, V)
    m = make_model(K, V, lam)
    m.update_variables(corpus(50, V, seed=2), max_iter=5)             # level 1 (default): updates only
    assert hip.trlda_model_last_merged(m._handle) == 0
    hip.trlda_model_set_merged_launch(m._handle, 2)
    m.update_variables(corpus(300, V, seed=1), max_iter=5)            # more documents than ride along
    assert hip.trlda_model_last_merged(m._handle) == 0
    m.update_variables(corpus(50, V, seed=2), max_iter=5)
    assert hip.trlda_model_last_merged(m._handle) == 1
    hip.trlda_model_set_sstats_mode(m._handle, 1)                     # atomic statistics
    m.update_variables(corpus(50, V, seed=2), max_iter=5)
    assert hip.trlda_model_last_merged(m._handle) == 0
    hip.trlda_model_set_sstats_mode(m._handle, 0)
    m7 = make_model(7, V, seeded_lambda(sampler, 7, 7, V))            # odd K: no pairs of topics
    hip.trlda_model_set_merged_launch(m7._handle, 2)
    m7.update_variables(corpus(50, V, seed=2), max_iter=5)
    assert hip.trlda_model_last_merged(m7._handle) == 0
    m200 = make_model(200, V, seeded_lambda(sampler, 7, 200, V))      # beyond the register kernel
    hip.trlda_model_set_merged_launch(m200._handle, 2)
    m200.update_variables(corpus(50, V, seed=2), max_iter=5)
    assert hip.trlda_model_last_merged(m200._handle) == 0


def test_repeated_ids_make_lists_longer_than_the_batch(hip, oracle, sampler):
    """ids may repeat within a document (reference src/lda.cpp:108 emits them, and
    nothing merges them), so a word's list can hold more entries than the batch has documents.
    The statistics stage of a merged launch walks a long list as sixteen chunks of at most 16
    entries; a batch of <= 256 documents with a list of more than 256 entries must not take it
    (round 4's condition looked at B only and dropped entries 16.. of every chunk).  V = 12,
    60 documents of 40..120 draws WITH replacement: lists of ~400 entries."""
    import trlda_amd
    from trlda_amd.documents import CSRDocuments
    K, V, B, D = 100, 12, 60, 5000
    rng = np.random.RandomState(4)
    lens = rng.randint(40, 121, size=B)
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = rng.randint(0, V, size=ip[-1]).astype(np.int32)
    cnts = rng.randint(0, 4, size=ip[-1]).astype(np.int32)
    docs = CSRDocuments(ip, ids, cnts)
    assert np.bincount(ids, minlength=V).max() > 256
    lam = seeded_lambda(sampler, 31, K, V)
    g0 = seeded_gamma(sampler, 32, K, B)
    go, so, ito = oracle.estep(lam, .1, ip, ids, cnts, g0, 20, 1e-3, nthreads=8)
    m = make_model(K, V, lam, D=D)
    for level in (2, 0):
        hip.trlda_model_set_merged_launch(m._handle, level)
        g, s, it = m.update_variables(docs, latents=g0, max_iter=20, threshold=1e-3, return_iterations=True)
        assert hip.trlda_model_last_merged(m._handle) == 0           # outside the stage's range
        assert np.array_equal(it, ito) and relerr(g, go) < TIGHT_RTOL
        check_sstats(s, so)
    # ... and a batch just inside: every list <= 256 entries although ids repeat
    lens2 = rng.randint(20, 41, size=B)
    ip2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int32)
    ids2 = rng.randint(0, V, size=ip2[-1]).astype(np.int32)
    cnts2 = rng.randint(1, 4, size=ip2[-1]).astype(np.int32)
    longest = np.bincount(ids2, minlength=V).max()
    assert B < longest <= 256
    docs2 = CSRDocuments(ip2, ids2, cnts2)
    go2, so2, ito2 = oracle.estep(lam, .1, ip2, ids2, cnts2, g0, 20, 1e-3, nthreads=8)
    hip.trlda_model_set_merged_launch(m._handle, 2)
    g, s, it = m.update_variables(docs2, latents=g0, max_iter=20, threshold=1e-3, return_iterations=True)
    assert hip.trlda_model_last_merged(m._handle) == 1
    assert np.array_equal(it, ito2) and relerr(g, go2) < TIGHT_RTOL
    check_sstats(s, so2)
    # the default update loop (merged level 1) on the long-list batch against the oracle
    lams = {}
    for level in (1, 0):
        mm = make_model(K, V, lam, D=D)
        hip.trlda_model_set_merged_launch(mm._handle, level)
        trlda_amd.seed(6)
        rho = mm.update_parameters(docs, max_iter_tr=3, max_iter_inference=20)
        lams[level] = (np.array(mm.lambdas), rho)
    oracle.seed(6)
    r_o, lam_o, _, _ = oracle.online_update_parameters(lam, .1, .3, D, ip, ids, cnts, 0, max_iter_tr=3,
                                                       max_iter_inference=20)
    assert lams[1][1] == lams[0][1] == r_o
    assert relerr(lams[1][0], lam_o) < 1e-9 and relerr(lams[0][0], lam_o) < 1e-9
    assert hip.trlda_model_synchronize(m._handle) == 0


@pytest.mark.parametrize("K,V,B", [(100, 7000, 200), (64, 900, 90)])
def test_update_loops_merged_against_the_kernels_of_their_own(hip, oracle, sampler, K, V, B):
    """OnlineLDA.update_parameters with and without the trust-region loop, three calls on three
    mini-batches: lambda from merged launches (one per trust-region iteration: statistics, M-step,
    row sums and the next exp(psi(lambda)) inside the document launch, the topic factors finished
    by workgroups of the NEXT launch) against the stand-alone kernels and against the oracle."""
    import trlda_amd
    D = 50000
    lam0 = seeded_lambda(sampler, 11, K, V)
    batches = [corpus(B, V, seed=300 + i, mean_unique=min(100, V // 8)) for i in range(3)]
    for tr in (4, 0):
        lams = {}
        for merged in (1, 0):
            m = make_model(K, V, lam0, D=D)
            hip.trlda_model_set_merged_launch(m._handle, merged)
            trlda_amd.seed(21)
            rhos = [m.update_parameters(b, max_iter_tr=tr, max_iter_inference=20) for b in batches]
            assert hip.trlda_model_last_merged(m._handle) == merged
            lams[merged] = (np.array(m.lambdas), rhos)
        assert lams[1][1] == lams[0][1]
        # (the row sums are added up in another order: a few ulp in the topic factors, carried
        # through twelve E-steps)
        assert relerr(lams[1][0], lams[0][0]) < 5e-11
        oracle.seed(21)
        lam, count = lam0, 0
        for b in batches:
            _, lam, _, _ = oracle.online_update_parameters(lam, .1, .3, D, b.indptr, b.ids, b.cnts, count,
                                                           max_iter_tr=tr, max_iter_inference=20)
            count += 1
        # (three calls, up to twelve E-steps deep: rounding differences grow along the trajectory
        # -- the stand-alone kernels are as far from the oracle, and 5e-11 from these)
        assert relerr(lams[1][0], lam) < 1e-8 and relerr(lams[0][0], lam) < 1e-8


def test_batch_lda_epochs_merged(hip, sampler):
    from trlda_amd.models import BatchLDA
    import trlda_amd
    K, V, B = 100, 2000, 150
    docs = corpus(B, V, seed=77)
    lam0 = seeded_lambda(sampler, 13, K, V)
    lams = {}
    for merged in (1, 0):
        m = BatchLDA(num_words=V, num_topics=K, alpha=.1, eta=.3)
        m.lambdas = lam0
        hip.trlda_model_set_merged_launch(m._handle, merged)
        trlda_amd.seed(5)
        m.update_parameters(docs, max_epochs=3, max_iter_inference=20)
        assert hip.trlda_model_last_merged(m._handle) == merged
        lams[merged] = np.array(m.lambdas)
    assert relerr(lams[1], lams[0]) < 5e-11


def test_many_merged_launches_in_a_row(hip, sampler):
    """600 consecutive merged launches on alternating batches without a synchronisation in
    between: the two counters only grow, every launch waits for ITS value."""
    K, V, B = 100, 7000, 200
    lam = seeded_lambda(sampler, 17, K, V)
    m = make_model(K, V, lam)
    hip.trlda_model_set_merged_launch(m._handle, 2)
    a, b = m.upload(corpus(B, V, seed=1, mean_unique=100)), m.upload(corpus(B - 9, V, seed=2, mean_unique=100))
    ga, gb = seeded_gamma(sampler, 18, K, B), seeded_gamma(sampler, 19, K, B - 9)
    first_a = m.update_variables(a, latents=ga, max_iter=20)
    first_b = m.update_variables(b, latents=gb, max_iter=20)
    for i in range(300):
        ra = m.update_variables(a, latents=ga, max_iter=20)
        rb = m.update_variables(b, latents=gb, max_iter=20)
    assert hip.trlda_model_last_merged(m._handle) == 1
    assert np.array_equal(ra[0], first_a[0]) and np.array_equal(ra[1], first_a[1])
    assert np.array_equal(rb[0], first_b[0]) and np.array_equal(rb[1], first_b[1])
    assert hip.trlda_model_synchronize(m._handle) == 0


@pytest.mark.parametrize("K,V,B", [(100, 3000, 2500), (200, 5000, 3300), (33, 800, 1100)])
def test_very_long_lists_are_cut_into_segments(hip, oracle, sampler, K, V, B):
    """csrc/estep_kernels.h, VeryLongArgs: a word with more than 1024 entries (here: words present
    in every one of 1100 .. 3300 documents, beside a Zipf vocabulary) is walked as segment tasks by
    whole workgroups and finished by whichever of them comes last -- against the one-workgroup-per-
    list form (to rounding: other partial sums), the oracle, run to run (bitwise), for plain
    E-steps and for update calls whose M-step rides on the statistics (even and odd K)."""
    import trlda_amd
    from trlda_amd.documents import CSRDocuments
    base = corpus(B, V, seed=7 + K, mean_unique=40)
    # three words in every document (ids no document has yet)
    ip, ids, cnts = [0], [], []
    for d in range(B):
        row = list(base.ids[base.indptr[d]:base.indptr[d + 1]])
        cn = list(base.cnts[base.indptr[d]:base.indptr[d + 1]])
        for w in (V - 1, V - 2, V - 3):
            if w not in row:
                row.append(w)
                cn.append(1 + (d + w) % 3)
        ids += row
        cnts += cn
        ip.append(len(ids))
    docs = CSRDocuments(np.array(ip, np.int32), np.array(ids, np.int32), np.array(cnts, np.int32))
    lam = seeded_lambda(sampler, 3, K, V)
    g0 = seeded_gamma(sampler, 4, K, B)
    m = make_model(K, V, lam)
    batch = m.upload(docs)
    assert hip.trlda_batch_num_very_long_words(batch.handle) >= 3
    res = {}
    for split in (1, 0):
        assert hip.trlda_model_set_split_lists(m._handle, split) == 0
        res[split] = m.update_variables(batch, latents=g0, max_iter=10, threshold=1e-3)
    again = m.update_variables(batch, latents=g0, max_iter=10, threshold=1e-3)
    hip.trlda_model_set_split_lists(m._handle, 1)
    once = m.update_variables(batch, latents=g0, max_iter=10, threshold=1e-3)
    twice = m.update_variables(batch, latents=g0, max_iter=10, threshold=1e-3)
    assert np.array_equal(once[1], twice[1]) and np.array_equal(once[1], res[1][1])
    assert np.array_equal(again[1], res[0][1])
    assert np.array_equal(res[1][0], res[0][0])
    assert relerr(res[1][1], res[0][1], floor=1e-200) < 1e-12
    go, so, ito = oracle.estep(lam, .1, docs.indptr, docs.ids, docs.cnts, g0, 10, 1e-3, nthreads=8)
    assert relerr(res[1][0], go) < TIGHT_RTOL
    check_sstats(res[1][1], so)
    # update calls: the M-step, the row sums and the next preamble ride on the same kernel
    D = 100000
    lams = {}
    for split in (1, 0):
        mm = make_model(K, V, lam, D=D)
        hip.trlda_model_set_split_lists(mm._handle, split)
        trlda_amd.seed(9)
        rho = [mm.update_parameters(batch, max_iter_tr=tr, max_iter_inference=10) for tr in (2, 0)]
        lams[split] = (np.array(mm.lambdas), rho)
    assert lams[1][1] == lams[0][1] and relerr(lams[1][0], lams[0][0]) < 1e-11
    oracle.seed(9)
    lo, count = lam, 0
    for tr in (2, 0):
        _, lo, _, _ = oracle.online_update_parameters(lo, .1, .3, D, docs.indptr, docs.ids, docs.cnts, count,
                                                       max_iter_tr=tr, max_iter_inference=10)
        count += 1
    assert relerr(lams[1][0], lo) < 1e-8
