"""Parity tests proper: the HIP path (through the C ABI) against the reference's golden
vectors and the pinned CPU oracle, on a real MI355X.

Bars: BASELINE.json's north star asks for gamma / lambda within 1e-5 relative (fp64); these
tests hold the kernels to TIGHT_RTOL = 1e-9, and iteration counts must be identical.
"""
import ctypes as C
import pickle

import numpy as np
import pytest

from helpers import (NORTH_STAR_RTOL, TIGHT_RTOL, HipSampler, golden, relerr, seeded_gamma,
                     seeded_lambda)

pytestmark = pytest.mark.gpu
ROOT_DIR = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip(hip_lib):
    from trlda_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a visible MI355X"
    return hip_lib


@pytest.fixture(scope="module")
def sampler(hip):
    return HipSampler(hip)


def make_model(K, V, lam, alpha=.1, eta=.3, D=1000):
    from trlda_amd.models import OnlineLDA
    m = OnlineLDA(num_words=V, num_topics=K, num_documents=D, alpha=alpha, eta=eta)
    m.lambdas = lam
    return m


def csr(f, suffix=""):
    from trlda_amd.documents import CSRDocuments
    return CSRDocuments(f["indptr" + suffix], f["ids" + suffix], f["cnts" + suffix])


def check_sstats(got, want, rtol=TIGHT_RTOL):
    assert relerr(got[want > 0], want[want > 0]) < rtol
    assert np.array_equal(got == 0, want == 0)


def test_native_library_is_loaded(hip):
    maps = open("/proc/self/maps").read()
    assert "libtrlda_hip.so" in maps


def test_device_digamma_table(hip):
    """The device psi, and exp(psi) in the log-free form the kernels use -- the general one, the
    call-free one for positive arguments and the one with the subtrahend of lda.cpp:173 -- against the
    reference's table (log grid 1e-6..1e6, small integers, reflection branch) and its three
    known answers (utils_test.py:33-51)."""
    f = golden("f0_rng_psi")
    x = np.ascontiguousarray(f["psi_x"])
    want = f["psi_y"]
    c = 1.75
    outs = [np.zeros_like(x) for _ in range(4)]
    rc = hip.trlda_debug_digamma(0, len(x), c, x.ctypes.data, *[o.ctypes.data for o in outs])
    assert rc == 0, hip.trlda_last_error()
    fin = np.isfinite(want)
    psi, epsi, epsi_lean, eminus = outs
    assert np.array_equal(np.isfinite(psi), fin)
    # psi crosses zero near x = 1.4616: use an absolute + relative bound
    err = np.abs(psi[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1.0)
    assert err.max() < 5e-15, (err.max(), x[fin][err.argmax()])
    with np.errstate(over="ignore", under="ignore"):
        ewant = np.exp(want)
        emwant = np.exp(want - c)
    ok = fin & (ewant > 1e-300) & (ewant < 1e300)
    for got, ref in ((epsi, ewant), (epsi_lean, ewant), (eminus, emwant)):
        # exp amplifies an absolute error in psi: the bound is relative to max(1, |psi|)
        rel = np.abs(got[ok] - ref[ok]) / ref[ok] / np.maximum(1.0, np.abs(want[ok]))
        assert rel.max() < 2e-15, (rel.max(), x[ok][rel.argmax()])
        assert (got[fin & (ref == 0)] == 0).all()
    # the call-free form for positive arguments (what the M-step kernels emit): the same operations
    # in the same order, except at the integers 1..10 (regular form instead of the exact harmonic
    # branch: a few ulp, inside the bound above)
    small_int = (x > 0) & (x <= 10) & (x == np.floor(x))
    assert small_int.any()
    assert np.array_equal(epsi[fin & ~small_int], epsi_lean[fin & ~small_int])
    # ... and at the ends of its range: 0 below 1e-290 like exp(-1/x), x itself from 1e25 on, NaN
    ends = np.array([1e-300, 5e-291, 1e25, 3e80, 1e200, np.inf, np.nan, 1.0, 2.0, 10.0])
    eo = [np.zeros_like(ends) for _ in range(4)]
    assert hip.trlda_debug_digamma(0, len(ends), 0., ends.ctypes.data, *[o.ctypes.data for o in eo]) == 0
    assert list(eo[2][:2]) == [0., 0.] and list(eo[2][2:6]) == list(ends[2:6]) and np.isnan(eo[2][6])
    assert np.max(np.abs(eo[2][7:] - eo[1][7:]) / eo[1][7:]) < 2e-15
    assert np.max(np.abs(eo[2][2:5] - eo[1][2:5]) / eo[1][2:5]) < 1e-13     # exp(log x) is the looser one
    kx = np.ascontiguousarray(f["kat_x"])
    ko = [np.zeros_like(kx) for _ in range(4)]
    assert hip.trlda_debug_digamma(0, len(kx), 0., kx.ctypes.data, *[o.ctypes.data for o in ko]) == 0
    assert np.max(np.abs(ko[0] - f["kat_y"])) < 1e-12
    for got in ko[1:]:
        assert np.max(np.abs(got - np.exp(f["kat_y"])) / np.exp(f["kat_y"])) < 1e-13


@pytest.mark.parametrize("name", ["f1a_estep", "f1b_estep"])
@pytest.mark.parametrize("mode", [0, 1])
def test_estep_golden(hip, sampler, name, mode):
    f = golden(name)
    K, V, B = int(f["K"]), int(f["V"]), int(f["B"])
    lam = seeded_lambda(sampler, f["lambda_seed"], K, V)
    g0 = seeded_gamma(sampler, f["gamma0_seed"], K, B)
    m = make_model(K, V, lam, alpha=f["alpha"])
    hip.trlda_model_set_sstats_mode(m._handle, mode)
    batch = m.upload(csr(f))
    for threads in (0, 64, 512):
        hip.trlda_model_set_doc_threads(m._handle, threads)
        for (it, thr) in [(0, 1e-3), (1, 1e-3), (20, 1e-3), (50, 0.0), (100, 1e-3)]:
            key = "it%d_thr%g" % (it, thr)
            g, s, iters = m.update_variables(batch, latents=g0, max_iter=it, threshold=thr,
                                             return_iterations=True)
            assert g.flags.f_contiguous and s.flags.f_contiguous
            assert g.shape == (K, B) and s.shape == (K, V)
            assert relerr(g, f["gamma_" + key]) < TIGHT_RTOL
            check_sstats(s, f["sstats_" + key])
            assert np.array_equal(iters, f["iters_" + key])


def test_estep_bench_shape_golden(hip, sampler):
    f = golden("f2_bench_shape")
    K, V, B = int(f["K"]), int(f["V"]), int(f["B"])
    lam = seeded_lambda(sampler, f["lambda_seed"], K, V)
    g0 = seeded_gamma(sampler, f["gamma0_seed"], K, B)
    m = make_model(K, V, lam)
    g, s = m.do_e_step(csr(f), latents=g0, max_iter=20)
    assert relerr(g, f["gamma"]) < TIGHT_RTOL
    check_sstats(s[:, f["active"]], f["sstats_active"])
    inactive = np.setdiff1d(np.arange(V), f["active"])
    assert (s[:, inactive] == 0).all()


@pytest.mark.parametrize("name", ["f3a_edge", "f3b_edge"])
@pytest.mark.parametrize("mode", [0, 1])
def test_estep_edge_documents(hip, sampler, name, mode):
    """empty document, duplicate ids, zero counts, single word, id = V-1 and 0, n_d > 64, a
    document longer than the LDS tile (streamed path in f3b), all-zero counts."""
    f = golden(name)
    K, V, B = int(f["K"]), int(f["V"]), int(f["B"])
    lam = seeded_lambda(sampler, f["lambda_seed"], K, V)
    g0 = seeded_gamma(sampler, f["gamma0_seed"], K, B)
    m = make_model(K, V, lam)
    hip.trlda_model_set_sstats_mode(m._handle, mode)
    for threads in (0, 128, 1024):
        hip.trlda_model_set_doc_threads(m._handle, threads)
        for (it, thr) in [(0, 1e-3), (30, 1e-3), (7, 0.0)]:
            key = "it%d_thr%g" % (it, thr)
            g, s = m.update_variables(csr(f), latents=g0, max_iter=it, threshold=thr)
            assert relerr(g, f["gamma_" + key]) < TIGHT_RTOL
            if "sstats_active_" + key in f:
                check_sstats(s[:, f["active"]], f["sstats_active_" + key])
            want_sum = float(f["sstats_sum_" + key])
            assert abs(s.sum() - want_sum) <= 1e-9 * max(1., abs(want_sum))


def test_hoffman_cross_implementation(hip):
    """The reference's own test_vi (onlinelda_test.py:39-68), list-of-tuples input."""
    f = golden("f7_hoffman_test_vi")
    m = make_model(int(f["K"]), int(f["V"]), f["lam"], D=int(f["B"]))
    docs = csr(f).to_list()
    g, s = m.do_e_step(docs, max_iter=50, latents=f["gamma0"])
    for tag in ("hoffman", "ref"):
        assert relerr(g, f["gamma_" + tag]) < TIGHT_RTOL
        check_sstats(s, f["sstats_" + tag])
    assert np.corrcoef(f["gamma_hoffman"].ravel(), g.ravel())[0, 1] > 0.99
    assert np.corrcoef(f["sstats_hoffman"].ravel(), s.ravel())[0, 1] > 0.99


@pytest.mark.parametrize("K,V,B,mean,long_lens", [(10, 1000, 700, 50, (190, 160, 145, 144, 129)),
                                                  (20, 3000, 400, 60, (191, 150, 150, 140, 133, 130, 129, 129)),
                                                  (32, 2000, 64, 40, (180, 130, 129)),
                                                  (16, 1500, 300, 50, (600, 400)),
                                                  (10, 1000, 900, 60, (1500, 700, 300, 260, 193, 192, 150)),
                                                  (20, 2500, 64, 30, (2500, 520))])
def test_a_wave_per_document_beside_long_documents(hip, oracle, K, V, B, mean, long_lens):
    """The wave-per-document form does not depend on EVERY document being short: the documents of more
    than 128 words (they lead the batch's sorted order) keep a workgroup each -- the tiered launch's
    register / single-orientation bodies -- and the workgroups behind them take eight short documents
    each (DocKernelArgs::small_block0).  Plain E-steps, an update loop (merged launches), early exits:
    against the oracle at 1e-9 with identical iteration counts, against the tiered launch without the
    form at 1e-12.  Long documents that are split over workgroups (the last cases: > 192 words) keep
    their segments' workgroups in front of the waves' (lda.cpp:174-204)."""
    import trlda_amd
    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    lens = np.clip(np.random.RandomState(K + B).poisson(mean, B), 1, 128)
    where = np.random.RandomState(B).choice(B, len(long_lens), replace=False)
    lens[where] = long_lens
    ip, ii, cc = make_corpus(B, V, seed=15 + K, mean_unique=mean, lengths=np.minimum(lens, V))
    docs = CSRDocuments(ip, ii, cc)
    lam = seeded_lambda(oracle, 41 + K, K, V)
    g0 = seeded_gamma(oracle, 42 + K, K, B)
    split_case = False                               # (split documents go with the form as well)
    out = {}
    for kind, name in ((3, "small"), (4, "reg")):
        m = make_model(K, V, lam, D=5000)
        _ffi.check(hip.trlda_model_set_doc_kernel(m._handle, kind))
        g, s, it = m.update_variables(docs, latents=g0, max_iter=60, return_iterations=True)
        used = hip.trlda_model_last_doc_kernel(m._handle)
        used = used.decode() if isinstance(used, bytes) else used
        if not split_case:
            assert ("small" in used) == (name == "small"), used
        trlda_amd.seed(78)
        m.update_parameters(docs, max_iter_tr=3, max_iter_inference=20)
        m.update_parameters(docs, max_iter_tr=0, max_iter_inference=20)
        out[name] = (g, s, it, m.lambdas, used)
        m.close()
    go, so, ito = oracle.estep(lam, .1, ip, ii, cc, g0, 60, 1e-3)
    g, s, it, lam_s, used = out["small"]
    assert np.array_equal(it, ito) and (it < 60).any()
    assert relerr(g, go) < TIGHT_RTOL
    check_sstats(s, so)
    gr, sr, itr, lam_r, _ = out["reg"]
    assert np.array_equal(it, itr) and relerr(g, gr) < 1e-12 and relerr(lam_s, lam_r) < 1e-9
    # the default: taken where the SHORT documents alone outnumber the CUs
    m = make_model(K, V, lam, D=5000)
    g_d, _, it_d = m.update_variables(docs, latents=g0, max_iter=60, return_iterations=True)
    used_d = hip.trlda_model_last_doc_kernel(m._handle)
    used_d = used_d.decode() if isinstance(used_d, bytes) else used_d
    if not split_case:
        assert ("small" in used_d) == (B - len(long_lens) > 256), (B, used_d)
    assert np.array_equal(it_d, it)
    m.close()


def test_goldens_through_the_wave_per_document_body(hip, sampler):
    """VERDICT r5 item 5's gate: the reference's own vectors f1a (K = 10), f1b (K = 20) and the Hoffman
    cross-check f7 (K = 20; onlinelda_test.py:39-68) through estep_docs_small_body, forced with
    trlda_model_set_doc_kernel(TRLDA_DOCS_SMALL), at 1e-9 with the golden iteration counts
    (lda.cpp:185-204)."""
    from trlda_amd import _ffi

    def used(m):
        u = hip.trlda_model_last_doc_kernel(m._handle)
        return u.decode() if isinstance(u, bytes) else u

    for name in ("f1a_estep", "f1b_estep"):
        f = golden(name)
        K, V, B = int(f["K"]), int(f["V"]), int(f["B"])
        lam = seeded_lambda(sampler, f["lambda_seed"], K, V)
        g0 = seeded_gamma(sampler, f["gamma0_seed"], K, B)
        m = make_model(K, V, lam, alpha=f["alpha"])
        _ffi.check(hip.trlda_model_set_doc_kernel(m._handle, 3))
        batch = m.upload(csr(f))
        for (it, thr) in [(0, 1e-3), (1, 1e-3), (20, 1e-3), (50, 0.0), (100, 1e-3)]:
            key = "it%d_thr%g" % (it, thr)
            g, s, iters = m.update_variables(batch, latents=g0, max_iter=it, threshold=thr,
                                             return_iterations=True)
            assert "small" in used(m), used(m)
            assert relerr(g, f["gamma_" + key]) < TIGHT_RTOL
            check_sstats(s, f["sstats_" + key])
            assert np.array_equal(iters, f["iters_" + key])
        m.close()
    f = golden("f7_hoffman_test_vi")
    m = make_model(int(f["K"]), int(f["V"]), f["lam"], D=int(f["B"]))
    _ffi.check(hip.trlda_model_set_doc_kernel(m._handle, 3))
    g, s = m.do_e_step(csr(f).to_list(), max_iter=50, latents=f["gamma0"])
    assert "small" in used(m), used(m)
    for tag in ("hoffman", "ref"):
        assert relerr(g, f["gamma_" + tag]) < TIGHT_RTOL
        check_sstats(s, f["sstats_" + tag])
    m.close()


def test_one_shot_c_abi(hip, oracle, sampler):
    """trlda_estep / trlda_mstep_blend / trlda_tr_init with plain host pointers."""
    from trlda_amd.utils.synthetic import make_corpus
    K, V, B, D = 9, 140, 17, 300
    ip, ii, cc = make_corpus(B, V, seed=8, mean_unique=30)
    lam = seeded_lambda(sampler, 21, K, V)
    g0 = seeded_gamma(sampler, 22, K, B)
    alpha = np.linspace(.05, .5, K)
    g = g0.copy(order="F")
    s = np.zeros((K, V), order="F")
    iters = np.zeros(B, np.int32)
    rc = hip.trlda_estep(K, V, B, ip, ii, cc, lam, alpha, g, s, 40, 1e-3, iters.ctypes.data, 0)
    assert rc == 0, hip.trlda_last_error()
    go, so, ito = oracle.estep(lam, alpha, ip, ii, cc, g0, 40, 1e-3)
    assert relerr(g, go) < TIGHT_RTOL
    check_sstats(s, so)
    assert np.array_equal(iters, ito)
    out = np.zeros((K, V), order="F")
    assert hip.trlda_mstep_blend(K, V, .3, .25, D / B, lam, s, out, 0) == 0
    assert relerr(out, oracle.mstep_blend(lam, so, .3, .25, D / B)) < TIGHT_RTOL
    assert hip.trlda_tr_init(K, V, B, D, .3, .25, ip, ii, cc, lam, out, 0) == 0
    assert relerr(out, oracle.tr_init(lam, ip, ii, cc, D, .3, .25)) < 1e-14
    # an out-of-range word id is an error, not UB
    bad = ii.copy()
    bad[3] = V
    assert hip.trlda_estep(K, V, B, ip, bad, cc, lam, alpha, g, s, 5, 1e-3, None, 0) == -3


def test_online_trajectories_golden(hip):
    """update_parameters under trlda_amd.seed(): TR in {0,3} x init_gamma x rho."""
    import trlda_amd
    from trlda_amd.models import OnlineLDA
    f = golden("f4_online_trajectory")
    K, V, D = int(f["K"]), int(f["V"]), int(f["D"])
    for case in range(int(f["num_cases"])):
        tr, init_gamma, rho, seed, count_want, r_empty = f["c%d_meta" % case]
        trlda_amd.seed(int(seed))
        m = OnlineLDA(num_words=V, num_topics=K, num_documents=D, alpha=.1, eta=.3)
        assert np.array_equal(m.lambdas, f["c%d_lambda0" % case])      # libc stream, bit-exact
        for i in range(3):
            r = m.update_parameters(csr(f, str(i)), max_iter_tr=int(tr), max_iter_inference=20,
                                    kappa=.7, tau=100., rho=float(rho),
                                    init_gamma=bool(init_gamma))
            assert abs(r - f["c%d_rhos" % case][i]) < 1e-15
            assert relerr(m.lambdas, f["c%d_lambda%d" % (case, i + 1)]) < TIGHT_RTOL
        before = m.lambdas
        assert m.update_parameters([], max_iter_tr=int(tr)) == 1.0      # onlinelda.cpp:54-56
        assert m.update_count == int(count_want) == 3
        assert np.array_equal(m.lambdas, before)


def test_empirical_bayes_and_adaptive_rate_golden(hip):
    """update_alpha / update_eta / adaptive (onlinelda.cpp:116-175) against trajectories of the
    compiled reference: alpha, eta, lambda and the returned rho after every call."""
    import ast
    import trlda_amd
    from trlda_amd.models import OnlineLDA
    f = golden("f8_empirical_bayes")
    K, V, D = int(f["K"]), int(f["V"]), int(f["D"])
    for case in range(int(f["num_cases"])):
        kw = {}
        for key, val in f["c%d_kwargs" % case]:
            kw[str(key)] = ast.literal_eval(str(val))
        trlda_amd.seed(3000 + case)
        m = OnlineLDA(num_words=V, num_topics=K, num_documents=D, alpha=np.linspace(.05, .4, K),
                      eta=.25)
        assert np.array_equal(m.lambdas, f["c%d_lambda0" % case])
        for i in range(4):
            args = dict(max_iter_tr=2, max_iter_inference=20, kappa=.7, tau=10., rho=-1.)
            args.update(kw)
            r = m.update_parameters(csr(f, str(i)), **args)
            assert abs(r - f["c%d_rhos" % case][i]) <= 1e-9 * abs(r), (case, i, kw)
            assert relerr(m.lambdas, f["c%d_lambda%d" % (case, i + 1)]) < 1e-8, (case, i, kw)
            assert relerr(m.alpha.ravel(), f["c%d_alpha%d" % (case, i + 1)]) < 1e-8, (case, i, kw)
            assert abs(m.eta - float(f["c%d_eta%d" % (case, i + 1)])) < 1e-8 * m.eta, (case, i, kw)
        assert m.update_count == 4


def _kwargs(arr):
    import ast
    return {str(k): ast.literal_eval(str(v)) for k, v in arr}


def test_batch_lda_line_searches_golden(hip):
    """BatchLDA with update_alpha / update_eta (batchlda.cpp:64-205) vs the compiled reference."""
    import trlda_amd
    from trlda_amd.models import BatchLDA
    f = golden("f9_batch_empirical_bayes")
    K, V = int(f["K"]), int(f["V"])
    for case in range(int(f["num_cases"])):
        kw = _kwargs(f["c%d_kwargs" % case])
        trlda_amd.seed(4000 + case)
        m = BatchLDA(num_words=V, num_topics=K, alpha=np.linspace(.1, .6, K), eta=.2)
        args = dict(max_epochs=3, max_iter_inference=50)
        args.update(kw)
        assert m.update_parameters(csr(f), **args) == 1.
        assert relerr(m.lambdas, f["c%d_lambda" % case]) < 1e-7, (case, kw)
        assert relerr(m.alpha.ravel(), f["c%d_alpha" % case]) < 1e-7, (case, kw)
        assert abs(m.eta - float(f["c%d_eta" % case])) < 1e-7 * m.eta, (case, kw)


def test_cumulative_lda_golden(hip):
    """CumulativeLDA (cumulativelda.cpp:49-153): lambda = lambda' + sstats from a re-drawn
    lambda each call, cumulative alpha statistics."""
    import pickle
    import trlda_amd
    from trlda_amd.models import CumulativeLDA
    f = golden("f10_cumulative")
    K, V = int(f["K"]), int(f["V"])
    for case in range(int(f["num_cases"])):
        kw = _kwargs(f["c%d_kwargs" % case])
        trlda_amd.seed(5000 + case)
        m = CumulativeLDA(num_words=V, num_topics=K, alpha=.15, eta=.3)
        assert (m.lambdas == .3).all()
        for i in range(3):
            args = dict(max_epochs=2, max_iter_inference=40)
            args.update(kw)
            assert m.update_parameters(csr(f, str(i)), **args) == 1.
            assert relerr(m.lambdas, f["c%d_lambda%d" % (case, i + 1)]) < 1e-7, (case, i, kw)
            assert relerr(m.alpha.ravel(), f["c%d_alpha%d" % (case, i + 1)]) < 1e-7, (case, i, kw)
        assert m.update_parameters([]) == 1.
        m2 = pickle.loads(pickle.dumps(m))
        assert np.array_equal(m2.lambdas, m.lambdas)


def test_lower_bound_golden(hip, oracle):
    """model.lower_bound (lda.cpp:297-360) on the reference's own test_lower_bound set-up and
    on a K > 64 case with a target num_documents: equal to the pinned oracle (which equals
    Hoffman's approx_bound, the reference test's yardstick, onlinelda_test.py:94-95), and
    within 1e-3 of the compiled reference, whose lda.cpp:334 reads a row where the word's
    column is meant."""
    import trlda_amd
    from trlda_amd.models import OnlineLDA
    f = golden("f11_lower_bound")
    m = OnlineLDA(num_words=int(f["V"]), num_topics=int(f["K"]), num_documents=int(f["D"]),
                  alpha=.1, eta=.3)
    m.lambdas = f["lam"]
    trlda_amd.seed(int(f["seed"]))
    got = m.lower_bound(csr(f))
    assert abs(got - float(f["elbo_oracle"])) < 1e-9 * abs(float(f["elbo_oracle"]))
    assert abs(got - float(f["elbo_hoffman"])) < 1e-9 * abs(float(f["elbo_hoffman"]))
    assert abs(got - float(f["elbo_ref"])) < 1e-3 * abs(float(f["elbo_ref"]))
    m2 = OnlineLDA(num_words=int(f["V2"]), num_topics=int(f["K2"]), num_documents=1000,
                   alpha=.1, eta=.3)
    m2.lambdas = f["lam2"]
    trlda_amd.seed(int(f["seed2"]))
    got2 = m2.lower_bound(csr(f, "2"), num_documents=int(f["num_documents2"]))
    assert abs(got2 - float(f["elbo_oracle2"])) < 1e-9 * abs(float(f["elbo_oracle2"]))
    # the default target of an OnlineLDA is its own num_documents (onlinelda.cpp:184-191)
    trlda_amd.seed(int(f["seed2"]))
    m2.num_documents = int(f["num_documents2"])
    assert m2.lower_bound(csr(f, "2")) == got2
    with pytest.raises(NotImplementedError):
        m2.lower_bound(csr(f, "2"), inference_method="gibbs")


def test_reference_readme_example(hip, tmp_path, monkeypatch):
    """The example of the reference's README.md:36-59 through the reference's own import path
    (`trlda`, served by trlda_amd): same imports, same constructor and update_parameters
    keywords, `data_train.dat` in the working directory, batches of 200, ten epochs."""
    from trlda_amd.utils.synthetic import csr_to_docs, make_corpus
    docs = csr_to_docs(*make_corpus(500, 7000, seed=3, mean_unique=60))
    (tmp_path / "data_train.dat").write_text(
        "".join("%d %s\n" % (len(d), " ".join("%d:%d" % t for t in d)) for d in docs))
    monkeypatch.chdir(tmp_path)

    from trlda.models import OnlineLDA
    from trlda.utils import load_documents

    # create model
    model = OnlineLDA(
        num_words=7000,
        num_topics=100,
        num_documents=1000000,
        alpha=.1,
        eta=.2)

    # train model for 10 epochs with a batch size of 200
    for epoch in range(10):
        for documents in load_documents('data_train.dat', 200):
            model.update_parameters(
                docs=documents,
                max_iter_tr=10,
                max_iter_inference=20,
                kappa=.7,
                tau=100.,
                update_alpha=True,
                update_eta=True)

    import trlda
    assert trlda.models.OnlineLDA is OnlineLDA and callable(trlda.seed)
    assert model.update_count == 10 * 3          # 500 lines: two full batches and the rest
    assert np.isfinite(model.lambdas).all() and (model.lambdas > 0).all()
    assert (model.alpha > 0).all() and model.eta > 0


def test_config1_golden(hip):
    """BASELINE.json configs[0] end to end: K=10, V=1000, 1k docs, batch 100, TR=10."""
    import trlda_amd
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.models import OnlineLDA
    from trlda_amd.utils.synthetic import make_corpus
    f = golden("f4b_config1")
    K, V, D, B = int(f["K"]), int(f["V"]), int(f["D"]), int(f["B"])
    docs = CSRDocuments(*make_corpus(D, V, seed=int(f["corpus_seed"]),
                                     mean_unique=int(f["mean_unique"])))
    trlda_amd.seed(int(f["seed"]))
    m = OnlineLDA(num_words=V, num_topics=K, num_documents=D)
    for b in range(D // B):
        r = m.update_parameters(docs.slice(b * B, (b + 1) * B), max_iter_tr=10,
                                max_iter_inference=20)
        assert abs(r - f["rhos"][b]) < 1e-15
    assert m.update_count == int(f["update_count"])
    assert relerr(m.lambdas, f["lambda_final"]) < 1e-7 < NORTH_STAR_RTOL


def test_batch_lda_golden(hip):
    import trlda_amd
    from trlda_amd.models import BatchLDA
    f = golden("f5_batch")
    trlda_amd.seed(int(f["seed"]))
    m = BatchLDA(num_words=int(f["V"]), num_topics=int(f["K"]), alpha=.1, eta=.3)
    assert np.array_equal(m.lambdas, f["lambda0"])
    assert m.update_parameters(csr(f), max_epochs=2, max_iter_inference=100) == 1.
    assert relerr(m.lambdas, f["lambda2"]) < TIGHT_RTOL
    assert m.update_parameters([]) == 1.


def test_default_gamma_comes_from_the_seeded_libc_stream(hip, oracle):
    import trlda_amd
    from trlda_amd.utils.synthetic import make_corpus
    K, V, B = 6, 80, 9
    ip, ii, cc = make_corpus(B, V, seed=5, mean_unique=20)
    lam = seeded_lambda(oracle, 9, K, V)
    m = make_model(K, V, lam, D=100)
    from trlda_amd.documents import CSRDocuments
    trlda_amd.seed(10)
    g, s = m.update_variables(CSRDocuments(ip, ii, cc), max_iter=15)
    g0 = seeded_gamma(oracle, 10, K, B)
    go, so, _ = oracle.estep(lam, .1, ip, ii, cc, g0, 15, 1e-3)
    assert relerr(g, go) < TIGHT_RTOL
    check_sstats(s, so)


# ---- BASELINE.json full sizes: size-independent properties ---------------------------

@pytest.fixture(scope="module")
def bench_case(hip, sampler):
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    K, V, B = 100, 7000, 200
    docs = CSRDocuments(*make_corpus(B, V, seed=20150707, mean_unique=100))
    lam = seeded_lambda(sampler, 1, K, V)
    g0 = seeded_gamma(sampler, 2, K, B)
    return K, V, B, docs, lam, g0


def test_full_size_invariants_and_oracle(hip, oracle, bench_case):
    K, V, B, docs, lam, g0 = bench_case
    m = make_model(K, V, lam, D=1000000)
    g, s, iters = m.update_variables(docs, latents=g0, max_iter=20, return_iterations=True)
    total = float(docs.cnts.sum())
    # sum sstats = sum counts; sum gamma = sum counts + B*sum(alpha)
    assert abs(s.sum() - total) < 1e-9 * total
    assert abs(g.sum() - (total + B * K * .1)) < 1e-9 * total
    assert (g > 0).all() and (s >= 0).all()
    go, so, ito = oracle.estep(lam, .1, docs.indptr, docs.ids, docs.cnts, g0, 20, 1e-3)
    assert relerr(g, go) < TIGHT_RTOL
    check_sstats(s, so)
    assert np.array_equal(iters, ito)


def test_full_size_bitwise_reproducible_and_mode_agreement(hip, bench_case):
    K, V, B, docs, lam, g0 = bench_case
    m = make_model(K, V, lam, D=1000000)
    batch = m.upload(docs)
    g1, s1 = m.update_variables(batch, latents=g0, max_iter=20)
    g2, s2 = m.update_variables(batch, latents=g0, max_iter=20)
    assert np.array_equal(g1, g2) and np.array_equal(s1, s2)      # segmented mode: bitwise
    hip.trlda_model_set_sstats_mode(m._handle, 1)
    g3, s3 = m.update_variables(batch, latents=g0, max_iter=20)
    assert np.array_equal(g1, g3)
    assert relerr(s3[s1 > 0], s1[s1 > 0]) < 1e-12                  # atomics: order only


def test_active_word_preamble_changes_nothing(hip, bench_case):
    """exp E[log beta] on the batch's words only vs on all V words: identical gamma / sstats."""
    K, V, B, docs, lam, g0 = bench_case
    m = make_model(K, V, lam, D=1000000)
    batch = m.upload(docs)
    g1, s1 = m.update_variables(batch, latents=g0, max_iter=20)
    assert hip.trlda_model_set_dense_preamble(m._handle, 1) == 0
    g2, s2 = m.update_variables(batch, latents=g0, max_iter=20)
    assert np.array_equal(g1, g2) and np.array_equal(s1, s2)
    for mode in (1,):
        hip.trlda_model_set_sstats_mode(m._handle, mode)
        hip.trlda_model_set_dense_preamble(m._handle, 0)
        g3, s3 = m.update_variables(batch, latents=g0, max_iter=20)
        assert np.array_equal(g1, g3) and np.isfinite(s3).all()
        assert relerr(s3[s1 > 0], s1[s1 > 0]) < 1e-12 and (s3[s1 == 0] == 0).all()


def test_full_size_document_permutation(hip, bench_case):
    """Documents are independent given lambda: permuting the batch permutes gamma and leaves
    the statistics unchanged up to summation order."""
    from trlda_amd.documents import CSRDocuments
    K, V, B, docs, lam, g0 = bench_case
    m = make_model(K, V, lam, D=1000000)
    g1, s1 = m.update_variables(docs, latents=g0, max_iter=20)
    perm = np.random.RandomState(0).permutation(B)
    lens = np.diff(docs.indptr)
    ip = np.zeros(B + 1, np.int32)
    ip[1:] = np.cumsum(lens[perm])
    ids = np.concatenate([docs.ids[docs.indptr[d]:docs.indptr[d + 1]] for d in perm])
    cnts = np.concatenate([docs.cnts[docs.indptr[d]:docs.indptr[d + 1]] for d in perm])
    g2, s2 = m.update_variables(CSRDocuments(ip, ids, cnts), latents=g0[:, perm], max_iter=20)
    assert np.array_equal(g2, g1[:, perm])
    assert relerr(s2[s1 > 0], s1[s1 > 0]) < 1e-12


@pytest.mark.parametrize("K", [100, 128, 7])
def test_document_length_boundaries(hip, oracle, sampler, K):
    """Documents right at the limits of the document kernels' tiers: 128 words (all in
    registers), 129..144 (18 words per wave), 145..192 (register part + LDS tail), 193 and up
    (single orientation: registers, then LDS rows -- 400 words -- then rows streamed from L2 --
    700 and 1300 words), all in ONE batch and one launch, with the fused preamble's topic factors
    applied by every variant; once more with everything forced through the general kernel."""
    from trlda_amd.documents import CSRDocuments
    V = 3000
    rng = np.random.RandomState(K)
    lam = seeded_lambda(sampler, 31, K, V)
    lens = [1, 2, 63, 64, 65, 127, 128, 129, 130, 137, 143, 144, 145, 150, 191, 192, 193, 250, 400,
            700, 1300]
    docs, ip = [], [0]
    for n in lens:
        ids = rng.permutation(V)[:n]
        cnts = 1 + rng.randint(3, size=n)
        docs.append((ids, cnts))
        ip.append(ip[-1] + n)
    ids = np.concatenate([d[0] for d in docs]).astype(np.int32)
    cnts = np.concatenate([d[1] for d in docs]).astype(np.int32)
    ip = np.array(ip, np.int32)
    g0 = seeded_gamma(sampler, 32, K, len(lens))
    m = make_model(K, V, lam)
    for kind in (0, 1):                          # TRLDA_DOCS_AUTO, TRLDA_DOCS_GENERAL
        assert hip.trlda_model_set_doc_kernel(m._handle, kind) == 0
        for (it, thr) in [(0, 0.), (1, 0.), (25, 1e-3)]:
            g, s, iters = m.update_variables(CSRDocuments(ip, ids, cnts), latents=g0, max_iter=it,
                                             threshold=thr, return_iterations=True)
            if kind == 0:
                assert hip.trlda_model_last_doc_kernel(m._handle) == b"estep_docs_tiered_kernel"
                assert hip.trlda_model_last_preamble_fused(m._handle) == 1
            go, so, ito = oracle.estep(lam, .1, ip, ids, cnts, g0, it, thr)
            per_doc = np.max(np.abs(g - go) / np.abs(go), axis=0)
            assert per_doc.max() < TIGHT_RTOL, list(zip(lens, per_doc))
            check_sstats(s, so)
            assert np.array_equal(iters, ito)


@pytest.mark.parametrize("K", [100, 128, 40])
def test_documents_split_over_workgroups(hip, oracle, sampler, K):
    """A document of 193 .. 2048 words is iterated by ceil(n / 128) workgroups that exchange K
    partial sums per iteration (estep_docs_reg_body<0, true>): same gamma, statistics and
    iteration counts as with one workgroup per document and as the oracle; lengths at the
    segment boundaries (193, 256, 257, 384, 385, 1024, 1025, 2048), early exits (threshold 1e-3
    over 60 iterations), both statistics modes; a batch that fills the chip with moderately long
    documents, or whose longest document is beyond the split range anyway, stays unsplit."""
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    V = 2600
    rng = np.random.RandomState(3 * K)
    lam = seeded_lambda(sampler, 71, K, V)
    lens = [5, 100, 128, 144, 192, 193, 256, 257, 384, 385, 600, 1024, 1025, 2048]
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = np.concatenate([rng.permutation(V)[:n] for n in lens]).astype(np.int32)
    cnts = (1 + rng.randint(4, size=ip[-1])).astype(np.int32)
    docs = CSRDocuments(ip, ids, cnts)
    g0 = seeded_gamma(sampler, 72, K, len(lens))
    m = make_model(K, V, lam)
    go, so, ito = oracle.estep(lam, .1, ip, ids, cnts, g0, 60, 1e-3)
    assert ito.min() < 60                          # some documents stop early
    want_wgs = sum(-(-n // 128) - 1 for n in lens if 192 < n <= 2048)
    for mode in (0, 1):
        hip.trlda_model_set_sstats_mode(m._handle, mode)
        res = {}
        for split in (1, 0):
            assert hip.trlda_model_set_split_docs(m._handle, split) == 0
            res[split] = m.update_variables(docs, latents=g0, max_iter=60, threshold=1e-3,
                                            return_iterations=True)
            assert hip.trlda_model_last_split_workgroups(m._handle) == (want_wgs if split else 0)
            assert hip.trlda_model_last_preamble_fused(m._handle) == 1
        (g1, s1, i1), (g2, s2, i2) = res[1], res[0]
        assert np.array_equal(i1, ito) and np.array_equal(i2, ito)
        per_doc = np.max(np.abs(g1 - go) / np.abs(go), axis=0)
        assert per_doc.max() < TIGHT_RTOL, list(zip(lens, per_doc))
        assert relerr(g1, g2) < 1e-11
        check_sstats(s1, so, rtol=TIGHT_RTOL if mode == 0 else 1e-8)
    hip.trlda_model_set_sstats_mode(m._handle, 0)
    hip.trlda_model_set_split_docs(m._handle, 1)
    # run to run: bitwise (the segments add the published rows in a fixed order)
    once = m.update_variables(docs, latents=g0, max_iter=60, threshold=1e-3)
    again = m.update_variables(docs, latents=g0, max_iter=60, threshold=1e-3)
    assert np.array_equal(again[0], once[0]) and np.array_equal(again[1], once[1])
    assert np.array_equal(once[0], res[1][0])        # gamma does not depend on the statistics mode
    # 300 documents of 200 words: 600 segments would take longer than 300 whole documents
    full = CSRDocuments(*make_corpus(300, V, seed=5, lengths=np.full(300, 200)))
    m.update_variables(full, max_iter=5)
    assert hip.trlda_model_last_split_workgroups(m._handle) == 0
    assert hip.trlda_model_last_doc_kernel(m._handle) == b"estep_docs_tiered_kernel"
    # a 2049-word document (one workgroup, rows in LDS and streamed) sets the launch's length
    ip3 = np.array([0, 2049, 2049 + 300], np.int32)
    ids3 = np.concatenate([rng.permutation(V)[:2049], rng.permutation(V)[:300]]).astype(np.int32)
    c3 = np.ones(len(ids3), np.int32)
    g3 = seeded_gamma(sampler, 73, K, 2)
    g, s_, it = m.update_variables(CSRDocuments(ip3, ids3, c3), latents=g3, max_iter=10,
                                   return_iterations=True)
    assert hip.trlda_model_last_split_workgroups(m._handle) == 0
    go3, so3, ito3 = oracle.estep(lam, .1, ip3, ids3, c3, g3, 10, 1e-3)
    assert relerr(g, go3) < TIGHT_RTOL and np.array_equal(it, ito3)


@pytest.mark.parametrize("K", [100, 128, 7])
@pytest.mark.parametrize("longest", [129, 137, 144])
def test_register_kernel_144_word_variant(hip, oracle, sampler, K, longest):
    """Batches whose longest document has 129..144 words run the register kernel's variant
    with 18 words per wave and a third register block (no LDS tail): document lengths around
    the wave boundaries (18, 36, ..) and the 128 / 144 limits, both statistics modes, fused and
    two-kernel preamble."""
    from trlda_amd.documents import CSRDocuments
    V = 2000
    rng = np.random.RandomState(100 * K + longest)
    lam = seeded_lambda(sampler, 61, K, V)
    lens = [0, 1, 17, 18, 19, 35, 36, 37, 125, 126, 127, 128, 129, 130, longest]
    docs, ip = [], [0]
    for n in lens:
        docs.append((rng.permutation(V)[:n], rng.randint(4, size=n)))
        ip.append(ip[-1] + n)
    ids = np.concatenate([d[0] for d in docs]).astype(np.int32)
    cnts = np.concatenate([d[1] for d in docs]).astype(np.int32)
    ip = np.array(ip, np.int32)
    g0 = seeded_gamma(sampler, 62, K, len(lens))
    m = make_model(K, V, lam)
    go, so, ito = oracle.estep(lam, .1, ip, ids, cnts, g0, 25, 1e-3)
    for mode in (0, 1):
        for split in (0, 1):
            hip.trlda_model_set_sstats_mode(m._handle, mode)
            hip.trlda_model_set_split_preamble(m._handle, split)
            g, s, iters = m.update_variables(CSRDocuments(ip, ids, cnts), latents=g0, max_iter=25,
                                             threshold=1e-3, return_iterations=True)
            # (one launch whose workgroups take the variant their own document needs)
            assert hip.trlda_model_last_doc_kernel(m._handle) == b"estep_docs_tiered_kernel"
            per_doc = np.max(np.abs(g - go) / np.abs(go), axis=0)
            assert per_doc.max() < TIGHT_RTOL, list(zip(lens, per_doc))
            check_sstats(s, so, rtol=TIGHT_RTOL if mode == 0 else 1e-8)
            assert np.array_equal(iters, ito)


def test_transposing_wave_reductions(hip):
    """The wave-level building block of the single-orientation document kernel
    (csrc/estep_wide.h): 16 (4, 2) sums over the 64 lanes in one butterfly, the result for
    value i delivered to the lanes whose bits 5,4,3,2 spell i."""
    rng = np.random.RandomState(7)
    x = rng.standard_normal((64, 16))
    outs = [np.zeros(64) for _ in range(3)]
    rc = hip.trlda_debug_fold16(0, np.ascontiguousarray(x).ctypes.data, *[o.ctypes.data for o in outs])
    assert rc == 0, hip.trlda_last_error()
    lane = np.arange(64)
    idx16 = ((lane >> 5) & 1) | ((lane >> 3) & 2) | ((lane >> 1) & 4) | ((lane << 1) & 8)
    idx4 = ((lane >> 5) & 1) | ((lane >> 3) & 2)
    idx2 = (lane >> 5) & 1
    tot = x.sum(axis=0)
    for got, idx in zip(outs, (idx16, idx4, idx2)):
        assert np.allclose(got, tot[idx], rtol=0, atol=1e-13), (got, tot[idx])


@pytest.mark.parametrize("K", [5, 64, 100, 129, 200, 257, 320, 333, 400, 500, 512])
@pytest.mark.parametrize("mode", [0, 1])
def test_single_orientation_kernel(hip, oracle, sampler, K, mode):
    """estep_docs_wide_kernel forced for every document: every slot count (ceil(K/64) = 1..8),
    documents that end inside the registers, inside the LDS rows and in the streamed part,
    the empty document, zero counts, max_iter 0 / 1 / until convergence."""
    from trlda_amd.documents import CSRDocuments
    V = 2500
    rng = np.random.RandomState(1000 + K)
    lam = seeded_lambda(sampler, 41, K, V)
    lens = [0, 1, 3, 7, 8, 9, 31, 64, 79, 80, 81, 100, 127, 160, 161, 200, 255, 256, 257, 300,
            420, 700, 1200]
    docs, ip = [], [0]
    for n in lens:
        ids = rng.permutation(V)[:n]
        cnts = rng.randint(4, size=n)            # zero counts are legal (onlinelda_test.py:57)
        docs.append((ids, cnts))
        ip.append(ip[-1] + n)
    ids = np.concatenate([d[0] for d in docs]).astype(np.int32)
    cnts = np.concatenate([d[1] for d in docs]).astype(np.int32)
    ip = np.array(ip, np.int32)
    g0 = seeded_gamma(sampler, 42, K, len(lens))
    m = make_model(K, V, lam)
    hip.trlda_model_set_sstats_mode(m._handle, mode)
    assert hip.trlda_model_set_doc_kernel(m._handle, 2) == 0
    for (it, thr) in [(0, 0.), (1, 0.), (30, 1e-3)]:
        g, s, iters = m.update_variables(CSRDocuments(ip, ids, cnts), latents=g0, max_iter=it,
                                         threshold=thr, return_iterations=True)
        go, so, ito = oracle.estep(lam, .1, ip, ids, cnts, g0, it, thr)
        per_doc = np.max(np.abs(g - go) / np.abs(go), axis=0)
        assert per_doc.max() < TIGHT_RTOL, list(zip(lens, per_doc))
        check_sstats(s, so, rtol=TIGHT_RTOL if mode == 0 else 1e-8)
        assert np.array_equal(iters, ito)


@pytest.mark.parametrize("K,V,lens", [
    (1, 50, [3, 0, 7, 50]), (2, 1, [1, 1, 0]), (3, 5, [0, 0, 0]), (513, 700, [5, 130, 300]),
    (600, 900, [1, 64, 200]), (100, 3000, [2500]), (128, 400, [400, 399, 1]),
    (129, 400, [400, 12]), (64, 2000, [1999, 1500, 3])])
def test_odd_shapes(hip, oracle, K, V, lens):
    """One topic, one word, a batch of empty documents, more than 512 topics (general kernel),
    a document holding most of the vocabulary: both statistics modes against the oracle."""
    from trlda_amd.documents import CSRDocuments
    rng = np.random.RandomState(3 + K)
    ip, ids, cnts = [0], [], []
    for n in lens:
        ids += list(rng.permutation(V)[:n])
        cnts += list(rng.randint(5, size=n))
        ip.append(ip[-1] + n)
    ip, ids, cnts = np.array(ip, np.int32), np.array(ids, np.int32), np.array(cnts, np.int32)
    hip.trlda_seed(K)
    lam = np.empty((K, V), order="F")
    hip.trlda_sample_gamma_init(K, V, lam)
    g0 = np.empty((K, len(lens)), order="F")
    hip.trlda_sample_gamma_init(K, len(lens), g0)
    m = make_model(K, V, lam)
    go, so, ito = oracle.estep(lam, .1, ip, ids, cnts, g0, 30, 1e-3)
    for mode in (0, 1):
        hip.trlda_model_set_sstats_mode(m._handle, mode)
        g, s, it = m.update_variables(CSRDocuments(ip, ids, cnts), latents=g0, max_iter=30,
                                      threshold=1e-3, return_iterations=True)
        assert relerr(g, go) < TIGHT_RTOL
        if (so > 0).any():
            check_sstats(s, so, rtol=TIGHT_RTOL if mode == 0 else 1e-8)
        else:
            assert not s.any()
        assert np.array_equal(it, ito)


@pytest.mark.parametrize("mode", [0, 1])
def test_fused_preamble_agrees_with_the_two_kernel_one(hip, oracle, sampler, mode):
    """Small tables run row sums + exp(psi(lambda)) as one launch and fold exp(-psiSum) into
    exp(psi(gamma)) in the document kernel (estep_kernels.h 2b): same gamma / sstats /
    iteration counts as the two-kernel preamble (a few ulp apart) and as the oracle; a batch
    with a long document keeps the fused preamble (every variant of the document launch applies
    the topic factors)."""
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    K, V, B = 100, 1500, 60
    ip, ii, cc = make_corpus(B, V, seed=21, mean_unique=70)
    lam = seeded_lambda(sampler, 51, K, V)
    g0 = seeded_gamma(sampler, 52, K, B)
    m = make_model(K, V, lam)
    hip.trlda_model_set_sstats_mode(m._handle, mode)
    docs = CSRDocuments(ip, ii, cc)
    out = {}
    for split in (0, 1):
        assert hip.trlda_model_set_split_preamble(m._handle, split) == 0
        out[split] = m.update_variables(docs, latents=g0, max_iter=40, threshold=1e-3,
                                        return_iterations=True)
        assert hip.trlda_model_last_preamble_fused(m._handle) == 1 - split
    (g1, s1, i1), (g2, s2, i2) = out[0], out[1]
    assert relerr(g1, g2) < 1e-10 and np.array_equal(i1, i2)     # ulps, 40 iterations on
    assert relerr(s1[s2 > 0], s2[s2 > 0]) < 1e-10 and np.array_equal(s1 == 0, s2 == 0)
    go, so, ito = oracle.estep(lam, .1, ip, ii, cc, g0, 40, 1e-3)
    assert relerr(g1, go) < TIGHT_RTOL and np.array_equal(i1, ito)
    check_sstats(s1, so, rtol=TIGHT_RTOL if mode == 0 else 1e-8)
    # the lower bound reads psiSum / the row sums the document kernel left behind
    import trlda_amd
    hip.trlda_model_set_split_preamble(m._handle, 0)
    trlda_amd.seed(9)
    lb_fused = m.lower_bound(docs)
    hip.trlda_model_set_split_preamble(m._handle, 1)
    trlda_amd.seed(9)
    lb_split = m.lower_bound(docs)
    assert abs(lb_fused - lb_split) < 1e-11 * abs(lb_split)
    # one document of 300 words: beyond the register variants, still one launch with the fused
    # preamble -- and the two-kernel form agrees
    ip2 = np.array([0, 300], np.int32)
    ids2 = np.arange(300, dtype=np.int32)
    long_doc = CSRDocuments(ip2, ids2, np.ones(300, np.int32))
    g0l = seeded_gamma(sampler, 53, K, 1)
    res = {}
    for split in (0, 1):
        hip.trlda_model_set_split_preamble(m._handle, split)
        res[split] = m.update_variables(long_doc, latents=g0l, max_iter=30, return_iterations=True)
        assert hip.trlda_model_last_preamble_fused(m._handle) == 1 - split
    assert relerr(res[0][0], res[1][0]) < 1e-10 and np.array_equal(res[0][2], res[1][2])
    gl, sl, itl = oracle.estep(lam, .1, ip2, ids2, np.ones(300, np.int32), g0l, 30, 1e-3)
    assert relerr(res[0][0], gl) < TIGHT_RTOL and np.array_equal(res[0][2], itl)
    check_sstats(res[0][1], sl, rtol=TIGHT_RTOL if mode == 0 else 1e-8)


@pytest.mark.parametrize("K,V,B", [(200, 50000, 48), (500, 100000, 40)])
def test_baseline_config_shapes(hip, oracle, K, V, B):
    """BASELINE.json's configs 4 and 5 at their full K and V (the single-orientation kernel,
    the wide row-sum grid with its combine kernel, 8 words per statistics workgroup at K=500),
    a small batch so that the oracle finishes in seconds; plus the size-independent
    invariants of SURVEY.md a17."""
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    ip, ii, cc = make_corpus(B, V, seed=2015 + K, mean_unique=100)
    rng = np.random.RandomState(K)                 # (the libc-stream sampler would need 5e9 draws)
    lam = np.asfortranarray(rng.gamma(100., .01, (K, V)))
    g0 = np.asfortranarray(rng.gamma(100., .01, (K, B)))
    m = make_model(K, V, lam)
    g, s, it = m.update_variables(CSRDocuments(ip, ii, cc), latents=g0, max_iter=20,
                                  threshold=1e-3, return_iterations=True)
    assert hip.trlda_model_last_doc_kernel(m._handle) == b"estep_docs_wide_kernel"
    go, so, ito = oracle.estep(lam, .1, ip, ii, cc, g0, 20, 1e-3, nthreads=8)
    assert relerr(g, go) < TIGHT_RTOL
    check_sstats(s, so)
    assert np.array_equal(it, ito)
    total = float(cc.sum())
    assert abs(s.sum() - total) < 1e-9 * total
    assert abs(g.sum() - (total + B * K * .1)) < 1e-9 * total


@pytest.mark.parametrize("lo,hi", [(-12, 2), (0, 12), (-8, 8)])
def test_extreme_values(hip, oracle, lo, hi):
    """lambda spread over many decades (exp E[log beta] down to underflow), tiny / integer
    gamma0 (the exact branches of psi), counts up to 1e6, document lengths across the kernel
    tiers.  Statistics below 1e-300 are left out: the device exp flushes denormals."""
    from trlda_amd.documents import CSRDocuments
    rng = np.random.RandomState(hi - lo)
    K, V = 100, 500
    ip, ids, cnts = [0], [], []
    for n in [5, 40, 100, 128, 130, 144, 150, 192, 200, 1, 0, 64]:
        ids += list(rng.permutation(V)[:n])
        cnts += list(rng.randint(1, 1000000, size=n))
        ip.append(ip[-1] + n)
    ip, ids, cnts = np.array(ip, np.int32), np.array(ids, np.int32), np.array(cnts, np.int32)
    B = len(ip) - 1
    lam = np.asfortranarray(10.0 ** rng.uniform(lo, hi, (K, V)))
    for g0 in (np.asfortranarray(rng.gamma(100, .01, (K, B))),
               np.asfortranarray(10.0 ** rng.uniform(-10, 0, (K, B))),
               np.asfortranarray(rng.randint(1, 11, (K, B)).astype(float))):
        m = make_model(K, V, lam, alpha=.01)
        g, s, it = m.update_variables(CSRDocuments(ip, ids, cnts), latents=g0, max_iter=30,
                                      threshold=1e-3, return_iterations=True)
        go, so, ito = oracle.estep(lam, .01, ip, ids, cnts, g0, 30, 1e-3)
        assert np.isfinite(g).all() and np.isfinite(s).all()
        assert relerr(g, go) < 1e-8
        big = so > 1e-300
        assert relerr(s[big], so[big]) < 1e-8
        assert np.array_equal(it, ito)


def test_converged_documents_stop_early(hip, oracle, sampler):
    """The data-dependent break (lda.cpp:202-203): iteration counts below max_iter, equal to
    the oracle's, per document."""
    from trlda_amd.utils.synthetic import make_corpus
    K, V, B = 8, 400, 64
    ip, ii, cc = make_corpus(B, V, seed=13, mean_unique=12)
    lam = seeded_lambda(sampler, 5, K, V)
    g0 = seeded_gamma(sampler, 6, K, B)
    m = make_model(K, V, lam)
    from trlda_amd.documents import CSRDocuments
    g, s, iters = m.update_variables(CSRDocuments(ip, ii, cc), latents=g0, max_iter=200,
                                     threshold=1e-3, return_iterations=True)
    go, so, ito = oracle.estep(lam, .1, ip, ii, cc, g0, 200, 1e-3)
    assert iters.max() < 200 and len(np.unique(iters)) > 1
    assert np.array_equal(iters, ito)
    assert relerr(g, go) < TIGHT_RTOL


# ---- the Python surface (onlinelda_test.py:14-35, 113-124, 176-200) ------------------

def test_basics_like_the_reference(hip):
    from trlda_amd.models import OnlineLDA
    W, D, K, alpha, eta = 102, 1010, 11, .27, 3.1
    model = OnlineLDA(num_words=W, num_topics=K, num_documents=D, alpha=alpha, eta=eta)
    assert model.num_topics == K and model.alpha.size == K
    assert model.num_documents == D and model.num_words == W
    assert model.alpha.ravel()[3] == alpha and model.alpha.shape == (K, 1)
    assert model.eta == eta
    with pytest.raises(RuntimeError):
        model.alpha = np.random.rand(K + 1)
    with pytest.raises(RuntimeError):
        model.alpha = -1.
    new_alpha = np.random.rand(K, 1)
    model.alpha = new_alpha
    assert np.max(np.abs(model.alpha.ravel() - new_alpha.ravel())) < 1e-20
    lam = model.lambdas
    assert lam.shape == (K, W) and lam.flags.f_contiguous and not lam.flags.writeable
    assert np.array_equal(model._lambda, lam)
    with pytest.raises(RuntimeError, match="Lambda has wrong dimensionality."):
        model.lambdas = np.ones((K, W + 1))
    model.lambdas = [[1.5] * W] * K                                  # nested lists are fine
    assert (model.lambdas == 1.5).all()
    with pytest.raises(RuntimeError):
        model.eta = -1.
    with pytest.raises(RuntimeError):
        model.num_documents = -5
    with pytest.raises(RuntimeError):
        model.update_count = -1
    with pytest.raises(RuntimeError, match="Initial gamma has wrong dimensionality."):
        model.update_variables([[(1, 1)]], latents=np.ones((K + 1, 1)))
    with pytest.raises(TypeError):
        model.update_variables([[(1, 1)]], inference_method="x")
    with pytest.raises(TypeError, match="Documents must be stored in a list."):
        model.update_parameters("nope")
    with pytest.raises(RuntimeError, match="word id"):
        model.update_variables([[(W, 1)]])
    assert "Number of topics: 11" in str(model)
    # an alpha array defines K (onlineldainterface.cpp:83)
    assert OnlineLDA(num_words=20, num_topics=10, num_documents=5, alpha=[.1, .1]).num_topics == 2


def test_m_step_counter(hip):
    from trlda_amd.models import OnlineLDA
    model = OnlineLDA(num_words=100, num_topics=10, num_documents=1000)
    model.update_parameters([])                                     # used to FPE in the reference
    docs = [[(int(w), 1) for w in np.random.permutation(100)[:5]] for _ in range(10)]
    model.update_parameters(docs)
    model.update_parameters(docs)
    assert model.update_count == 2


def test_pickle(hip):
    from trlda_amd.models import BatchLDA, OnlineLDA
    model0 = OnlineLDA(num_words=300, num_topics=50, num_documents=11110, alpha=np.random.rand(),
                       eta=np.random.rand())
    model0.update_count = 7
    model1 = pickle.loads(pickle.dumps({'model': model0}))['model']
    assert model0.num_words == model1.num_words and model0.num_topics == model1.num_topics
    assert model0.num_documents == model1.num_documents and model1.update_count == 7
    assert np.max(np.abs(model0.lambdas - model1.lambdas)) < 1e-20
    assert np.max(np.abs(model0.alpha - model1.alpha)) < 1e-20
    assert abs(model0.eta - model1.eta) < 1e-20
    b0 = BatchLDA(num_words=40, num_topics=5, alpha=.2, eta=.4)
    b1 = pickle.loads(pickle.dumps(b0))
    assert np.array_equal(b0.lambdas, b1.lambdas) and b1.eta == .4


def test_sharded_model_single_rank_matches_online_lda(hip):
    """trlda_amd.distributed with the real HIP engine (world size 1, no collective)."""
    import trlda_amd
    from trlda_amd.distributed import ShardedOnlineLDA
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.models import OnlineLDA
    from trlda_amd.utils.synthetic import make_corpus
    K, V, D, B = 12, 300, 2000, 40
    docs = [CSRDocuments(*make_corpus(B, V, seed=60 + i, mean_unique=30)) for i in range(2)]
    trlda_amd.seed(3)
    a = OnlineLDA(num_words=V, num_topics=K, num_documents=D)
    ra = [a.update_parameters(d, max_iter_tr=3) for d in docs]
    trlda_amd.seed(3)
    b = ShardedOnlineLDA(V, K, D, device=0)
    rb = [b.update_parameters(d, max_iter_tr=3) for d in docs]
    assert ra == rb and b.update_count == 2
    # (the model fuses the M-step into the statistics kernel, the composition path blends in a
    # kernel of its own: equal to rounding, not bit for bit)
    assert relerr(a.lambdas, b.lambdas) < 1e-10


def test_sharded_model_through_rccl_world_1(hip, tmp_path):
    """The multi-GPU composition end to end on the one GPU there is: a 1-rank NCCL (= RCCL)
    process group, HipEngine, the word-count and statistics all-reduces on device tensors.
    Run in a child process so that the process group does not leak into this one."""
    import subprocess
    import sys
    script = tmp_path / "rccl1.py"
    script.write_text("""
import os, sys
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
import numpy as np, torch, torch.distributed as dist
import trlda_amd
from trlda_amd.distributed import ShardedOnlineLDA
from trlda_amd.documents import CSRDocuments
from trlda_amd.models import OnlineLDA
from trlda_amd.utils.synthetic import make_corpus
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
K, V, D, B = 12, 300, 2000, 40
docs = [CSRDocuments(*make_corpus(B, V, seed=60 + i, mean_unique=30)) for i in range(2)]
trlda_amd.seed(3)
a = OnlineLDA(num_words=V, num_topics=K, num_documents=D)
ra = [a.update_parameters(d, max_iter_tr=3) for d in docs]
trlda_amd.seed(3)
b = ShardedOnlineLDA(V, K, D, device=0, exchange="sstats")
assert b.world == 1 and dist.is_initialized()
b._all_reduce_calls = 0
orig = b._all_reduce
def counted(t):
    b._all_reduce_calls += 1
    dist.all_reduce(t)            # world 1: identity, but through RCCL on the device tensor
    return t
b._all_reduce = counted
rb = [b.update_parameters(d, max_iter_tr=3) for d in docs]
assert ra == rb and b.update_count == 2
assert b._all_reduce_calls == 2 * (1 + 3)      # word counts + one per trust-region E-step
assert float(np.max(np.abs(a.lambdas - b.lambdas) / b.lambdas)) < 1e-10   # fused vs composed M-step
# the factor exchange (the default where it moves fewer bytes): the single-GPU call with the
# statistics as a kernel of their own (as the *_dp calls have them: an exchange sits in between),
# bit for bit; with the statistics inside the document launch (csrc/estep_merged.h) to rounding
trlda_amd.seed(3)
c = ShardedOnlineLDA(V, K, D, device=0)
assert c.use_factors(docs[0], docs[0].shard_cuts(1))
rc = [c.update_parameters(d, max_iter_tr=3) for d in docs]
assert rc == ra and c.update_count == 2
assert float(np.max(np.abs(a.lambdas - c.lambdas) / c.lambdas)) < 1e-11
from trlda_amd import _ffi
trlda_amd.seed(3)
a2 = OnlineLDA(num_words=V, num_topics=K, num_documents=D)
_ffi.check(_ffi.lib().trlda_model_set_merged_launch(a2._handle, 0))
assert [a2.update_parameters(d, max_iter_tr=3) for d in docs] == ra
assert np.array_equal(a2.lambdas, c.lambdas)
dist.destroy_process_group()
print("RCCL1-OK")
""" % ROOT_DIR)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert "RCCL1-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("K,V,B,mean", [(10, 1000, 100, 60), (20, 300, 64, 40), (32, 2000, 37, 100), (2, 50, 9, 6),
                                        (16, 500, 200, 120), (10, 1000, 700, 50),
                                        (7, 400, 300, 25), (15, 800, 64, 40), (31, 2000, 300, 60)])   # (odd K: 8-byte row loads)
def test_a_wave_per_document_at_small_k(hip, oracle, K, V, B, mean):
    """Round 6: at K <= 32 a document is one WAVE -- rows in registers, the sums over
    the words by a transposing butterfly, no LDS or barrier inside the fixed point
    (estep_docs_small_body; lda.cpp:174-204).  Against the oracle at 1e-9 with identical iteration
    counts, against the workgroup-per-document body (TRLDA_DOCS_REG) at 1e-12 with identical
    counts, with early exits (max_iter 60), through plain E-steps, an update loop (merged launches)
    and a fixed-lambda stream (deferred statistics, two lanes)."""
    import trlda_amd
    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    lens = np.clip(np.random.RandomState(K).poisson(mean, B), 0, 128)
    lens[:3] = [0, 1, 128]
    ip, ii, cc = make_corpus(B, V, seed=5 + K, mean_unique=mean, lengths=np.minimum(lens, V))
    docs = CSRDocuments(ip, ii, cc)
    lam = seeded_lambda(oracle, 31 + K, K, V)
    g0 = seeded_gamma(oracle, 32 + K, K, B)
    out = {}
    for kind, name in ((3, "small"), (4, "reg")):
        m = make_model(K, V, lam, D=5000)
        _ffi.check(hip.trlda_model_set_doc_kernel(m._handle, kind))
        g, s, it = m.update_variables(docs, latents=g0, max_iter=60, return_iterations=True)
        used = hip.trlda_model_last_doc_kernel(m._handle)
        used = used.decode() if isinstance(used, bytes) else used
        assert ("small" in used) == (name == "small"), used
        trlda_amd.seed(77)
        m.update_parameters(docs, max_iter_tr=3, max_iter_inference=20)
        m.update_parameters(docs, max_iter_tr=0, max_iter_inference=20)
        out[name] = (g, s, it, m.lambdas)
        m.close()
    go, so, ito = oracle.estep(lam, .1, ip, ii, cc, g0, 60, 1e-3)
    g, s, it, lam_s = out["small"]
    assert np.array_equal(it, ito) and (it < 60).any()
    assert relerr(g, go) < TIGHT_RTOL
    check_sstats(s, so)
    gr, sr, itr, lam_r = out["reg"]
    assert np.array_equal(it, itr) and relerr(g, gr) < 1e-12 and relerr(lam_s, lam_r) < 1e-9
    # the default: a wave per document where the batch has more documents than the device has CUs
    m = make_model(K, V, lam, D=5000)
    g_d, _, it_d = m.update_variables(docs, latents=g0, max_iter=60, return_iterations=True)
    used = hip.trlda_model_last_doc_kernel(m._handle)
    used = used.decode() if isinstance(used, bytes) else used
    assert ("small" in used) == (B > 256), (B, used)
    assert np.array_equal(g_d, g if B > 256 else gr) and np.array_equal(it_d, it)
    m.close()
    nz = sr > 0
    assert relerr(s[nz], sr[nz]) < 1e-11 and np.array_equal(s == 0, sr == 0)
