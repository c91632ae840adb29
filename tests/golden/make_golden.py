"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the build container only).

    python tests/golden/make_golden.py

Sources of truth, none of which travel to the GPU box:
  * oracle/_ref/libtrlda_ref.so -- the reference's unmodified C++ core
    (/root/reference/code/trlda/src/*.cpp) behind oracle/ref_shim.cpp, built by
    oracle/Makefile;
  * /root/reference/code/trlda/python/tests/onlineldavb.py -- M. Hoffman's NumPy
    online LDA, the cross-implementation oracle of the reference's own test_vi
    (onlinelda_test.py:39-68); imported here, never copied (GPL-3).

The fixtures are DATA only: inputs (or the srand() seed that reproduces them through the
libc-rand() sampleGamma stream) and the reference's outputs.  Where lambda would be large
it is stored as a seed: `srand(seed); sampleGamma(K, V, 100) / 100` is bit-reproducible
(F0 checks that against the reference).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from trlda_amd.utils.synthetic import make_corpus  # noqa: E402

REF_TESTS = "/root/reference/code/trlda/python/tests"


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f kB" % (name + ".npz", os.path.getsize(path) / 1e3))


def edge_corpus(V, rng):
    """Edge documents of SURVEY.md 8c (F3)."""
    docs = [
        [],                                                   # empty document
        [(3, 2), (3, 1), (7, 4), (3, 5)],                     # duplicate ids
        [(1, 0), (2, 0), (5, 3)],                             # zero counts
        [(V - 1, 1)],                                         # single word, id = V-1
        [(0, 7)],                                             # id = 0
        [(int(w), int(1 + rng.integers(4))) for w in rng.permutation(V)[:70]],   # n_d > 64
        [(int(w), int(1 + rng.integers(4))) for w in rng.permutation(V)[:min(V, 260)]],  # > LDS tile
        [(int(w), 0) for w in rng.permutation(V)[:5]],        # all counts zero
    ]
    n = np.array([len(d) for d in docs])
    indptr = np.zeros(len(docs) + 1, np.int32)
    indptr[1:] = np.cumsum(n)
    flat = np.array([t for d in docs for t in d], dtype=np.int32).reshape(-1, 2)
    return indptr, flat[:, 0].copy(), flat[:, 1].copy()


def main():
    pyoracle.build(ref=True)
    ref = pyoracle.Reference()
    orc = pyoracle.Oracle()

    # ---- F0: sampleGamma stream + F6: psi table ---------------------------------------
    ref.seed(42)
    sg = ref.sample_gamma(3, 2, 100)
    ref.seed(7)
    sg2 = ref.sample_gamma(5, 4, 3)
    x = np.concatenate([np.logspace(-6, 6, 241), np.arange(1, 13, dtype=float),
                        [0.1, 0.5, 9.999999, 10.000001, 1e17, 2e17, -0.5, -1.5, -2.25, -0.75]])
    psi = np.array([ref.digamma(v) for v in x])
    # the reference's own known-answer values (utils_test.py:33-51, rows with n = 0)
    kat_x = np.array([0.1, 1.0, 120.0])
    kat_y = np.array([-10.423754940411, -0.5772156649015329, 4.7833192891185])
    save("f0_rng_psi", sg_seed42_3x2x100=sg, sg_seed7_5x4x3=sg2, psi_x=x, psi_y=psi,
         kat_x=kat_x, kat_y=kat_y)

    # ---- F1: do_e_step parity, full inputs -------------------------------------------
    for tag, (K, V, B, mu) in {"a": (10, 1000, 100, 40), "b": (20, 300, 16, 60)}.items():
        indptr, ids, cnts = make_corpus(B, V, seed=20150706, mean_unique=mu)
        ref.seed(101)
        m = ref.online(V, K, 1000, alpha=.1, eta=.3)          # draws lambda from rand()
        lam = m.lambdas
        alpha = np.linspace(0.05, 0.3, K) if tag == "b" else np.full(K, .1)
        m.alpha = alpha
        ref.seed(202)
        g0 = ref.sample_gamma(K, B, 100) / 100.
        out = dict(K=K, V=V, B=B, indptr=indptr, ids=ids, cnts=cnts, lambda_seed=101,
                   gamma0_seed=202, alpha=alpha)
        if tag == "b":
            out["lam"] = lam                                  # one fixture with lambda stored
        for (it, thr) in [(0, 1e-3), (1, 1e-3), (20, 1e-3), (50, 0.0), (100, 1e-3)]:
            g, s = m.estep(indptr, ids, cnts, g0, it, thr)
            _, _, iters = orc.estep(lam, alpha, indptr, ids, cnts, g0, it, thr)
            key = "it%d_thr%g" % (it, thr)
            out["gamma_" + key] = g
            out["sstats_" + key] = s
            out["iters_" + key] = iters
        save("f1%s_estep" % tag, **out)

    # ---- F2: bench shape K=100, V=7000, B=8; outputs on active columns ---------------
    K, V, B = 100, 7000, 8
    indptr, ids, cnts = make_corpus(B, V, seed=20150707, mean_unique=100)
    ref.seed(1)
    m = ref.online(V, K, 1000000, alpha=.1, eta=.3)
    ref.seed(2)
    g0 = ref.sample_gamma(K, B, 100) / 100.
    g, s = m.estep(indptr, ids, cnts, g0, 20, 1e-3)
    active = np.unique(ids)
    save("f2_bench_shape", K=K, V=V, B=B, indptr=indptr, ids=ids, cnts=cnts, lambda_seed=1,
         gamma0_seed=2, gamma=g, active=active, sstats_active=s[:, active],
         sstats_sum=s.sum(), gamma_sum=g.sum())

    # ---- F3: edge documents (two K: all staged in LDS / some streamed) ---------------
    for tag, (K, V) in {"a": (12, 300), "b": (160, 400)}.items():
        rng = np.random.Generator(np.random.PCG64(5))
        indptr, ids, cnts = edge_corpus(V, rng)
        B = len(indptr) - 1
        ref.seed(11)
        m = ref.online(V, K, 500, alpha=.1, eta=.3)
        ref.seed(12)
        g0 = ref.sample_gamma(K, B, 100) / 100.
        out = dict(K=K, V=V, B=B, indptr=indptr, ids=ids, cnts=cnts, lambda_seed=11,
                   gamma0_seed=12)
        for (it, thr) in [(0, 1e-3), (30, 1e-3), (7, 0.0)]:
            g, s = m.estep(indptr, ids, cnts, g0, it, thr)
            key = "it%d_thr%g" % (it, thr)
            active = np.unique(ids)
            out["gamma_" + key] = g
            if it > 0 or tag == "a":
                out["sstats_active_" + key] = s[:, active]
            out["sstats_sum_" + key] = s.sum()
        out["active"] = np.unique(ids)
        save("f3%s_edge" % tag, **out)

    # ---- F4: OnlineLDA.update_parameters trajectories ---------------------------------
    K, V, D, B = 10, 200, 1000, 25
    batches = [make_corpus(B, V, seed=20150706 + i, mean_unique=30) for i in range(3)]
    out = dict(K=K, V=V, D=D, B=B)
    for i, (ip, ii, cc) in enumerate(batches):
        out["indptr%d" % i], out["ids%d" % i], out["cnts%d" % i] = ip, ii, cc
    case = 0
    for tr in (0, 3):
        for init_gamma in (True, False):
            for rho in (-1., 0.1):
                ref.seed(1000 + case)
                m = ref.online(V, K, D, alpha=.1, eta=.3)
                out["c%d_lambda0" % case] = m.lambdas
                rhos = []
                for i, (ip, ii, cc) in enumerate(batches):
                    r = m.update_parameters(ip, ii, cc, max_iter_tr=tr, max_iter_inference=20,
                                            kappa=.7, tau=100., rho=rho, init_gamma=init_gamma)
                    rhos.append(r)
                    out["c%d_lambda%d" % (case, i + 1)] = m.lambdas
                # an empty batch neither counts nor changes lambda (onlinelda.cpp:54-56)
                r_empty = m.update_parameters(np.zeros(1, np.int32), np.zeros(0, np.int32),
                                              np.zeros(0, np.int32), max_iter_tr=tr)
                out["c%d_meta" % case] = np.array([tr, int(init_gamma), rho, 1000 + case,
                                                   m.update_count, r_empty])
                out["c%d_rhos" % case] = np.array(rhos)
                case += 1
    out["num_cases"] = case
    save("f4_online_trajectory", **out)

    # config 1 of BASELINE.json: K=10, V=1000, 1k docs, batch 100, TR=10, 20 inner iterations
    K, V, D, B = 10, 1000, 1000, 100
    ip, ii, cc = make_corpus(D, V, seed=20150706, mean_unique=50)
    ref.seed(77)
    m = ref.online(V, K, D, alpha=.1, eta=.3)
    rhos = []
    for b in range(D // B):
        lo, hi = ip[b * B], ip[(b + 1) * B]
        rhos.append(m.update_parameters(ip[b * B:(b + 1) * B + 1] - lo, ii[lo:hi], cc[lo:hi],
                                        max_iter_tr=10, max_iter_inference=20))
    save("f4b_config1", K=K, V=V, D=D, B=B, corpus_seed=20150706, mean_unique=50, seed=77,
         rhos=np.array(rhos), lambda_final=m.lambdas, update_count=m.update_count)

    # ---- F8: empirical-Bayes alpha / eta and the adaptive learning rate ------------------
    K, V, D, B = 8, 120, 800, 30
    batches = [make_corpus(B, V, seed=20150800 + i, mean_unique=25) for i in range(4)]
    out = dict(K=K, V=V, D=D, B=B)
    for i, (ip, ii, cc) in enumerate(batches):
        out["indptr%d" % i], out["ids%d" % i], out["cnts%d" % i] = ip, ii, cc
    cases = [dict(update_alpha=True), dict(update_eta=True), dict(update_alpha=True, update_eta=True),
             dict(adaptive=True), dict(adaptive=True, update_alpha=True, update_eta=True),
             dict(update_alpha=True, update_lambda=False, rho=.1),
             dict(update_alpha=True, update_eta=True, max_iter_tr=0)]
    for c, kw in enumerate(cases):
        ref.seed(3000 + c)
        m = ref.online(V, K, D, alpha=np.linspace(.05, .4, K), eta=.25)
        out["c%d_lambda0" % c] = m.lambdas
        rhos = []
        for i, (ip, ii, cc) in enumerate(batches):
            args = dict(max_iter_tr=2, max_iter_inference=20, kappa=.7, tau=10., rho=-1.)
            args.update(kw)
            rhos.append(m.update_parameters(ip, ii, cc, **args))
            out["c%d_lambda%d" % (c, i + 1)] = m.lambdas
            out["c%d_alpha%d" % (c, i + 1)] = m.alpha
            out["c%d_eta%d" % (c, i + 1)] = np.array(m.ref.lib.ref_model_get_eta(m.h))
        out["c%d_rhos" % c] = np.array(rhos)
        out["c%d_kwargs" % c] = np.array(sorted(kw.items()), dtype=object).astype(str)
    out["num_cases"] = len(cases)
    save("f8_empirical_bayes", **out)

    # ---- F9: BatchLDA with the alpha / eta line searches; F10: CumulativeLDA ------------
    K, V, B = 6, 90, 35
    ip, ii, cc = make_corpus(B, V, seed=20150900, mean_unique=20)
    out = dict(K=K, V=V, B=B, indptr=ip, ids=ii, cnts=cc)
    cases = [dict(update_alpha=True), dict(update_eta=True), dict(update_alpha=True, update_eta=True),
             dict(update_alpha=True, update_lambda=False)]
    for c, kw in enumerate(cases):
        ref.seed(4000 + c)
        m = ref.batch(V, K, alpha=np.linspace(.1, .6, K), eta=.2)
        args = dict(max_epochs=3, max_iter_inference=50)
        args.update(kw)
        m.update_parameters(ip, ii, cc, **args)
        out["c%d_lambda" % c], out["c%d_alpha" % c], out["c%d_eta" % c] = m.lambdas, m.alpha, np.array(m.eta)
        out["c%d_kwargs" % c] = np.array(sorted(kw.items()), dtype=object).astype(str)
    out["num_cases"] = len(cases)
    save("f9_batch_empirical_bayes", **out)

    K, V, B = 7, 110, 30
    batches = [make_corpus(B, V, seed=20151000 + i, mean_unique=22) for i in range(3)]
    out = dict(K=K, V=V, B=B)
    for i, (ip, ii, cc) in enumerate(batches):
        out["indptr%d" % i], out["ids%d" % i], out["cnts%d" % i] = ip, ii, cc
    cases = [dict(), dict(update_alpha=True), dict(update_alpha=True, update_lambda=False),
             dict(max_epochs=1, inference_threshold=0.)]
    for c, kw in enumerate(cases):
        ref.seed(5000 + c)
        m = ref.cumulative(V, K, alpha=.15, eta=.3)
        for i, (ip, ii, cc) in enumerate(batches):
            args = dict(max_epochs=2, max_iter_inference=40)
            args.update(kw)
            m.update_parameters(ip, ii, cc, **args)
            out["c%d_lambda%d" % (c, i + 1)] = m.lambdas
            out["c%d_alpha%d" % (c, i + 1)] = m.alpha
        out["c%d_kwargs" % c] = np.array(sorted(kw.items()), dtype=object).astype(str) \
            if kw else np.zeros((0, 2), dtype=str)
    out["num_cases"] = len(cases)
    save("f10_cumulative", **out)

    # ---- F5: BatchLDA, 2 epochs --------------------------------------------------------
    K, V, B = 8, 150, 40
    ip, ii, cc = make_corpus(B, V, seed=20150709, mean_unique=25)
    ref.seed(31)
    m = ref.batch(V, K, alpha=.1, eta=.3)
    lam0 = m.lambdas
    m.update_parameters(ip, ii, cc, max_epochs=2, max_iter_inference=100)
    save("f5_batch", K=K, V=V, B=B, indptr=ip, ids=ii, cnts=cc, seed=31, lambda0=lam0,
         lambda2=m.lambdas)

    # ---- F7: Hoffman's onlineldavb on the reference's own test_vi set-up ----------------
    sys.path.insert(0, REF_TESTS)
    import onlineldavb  # noqa: E402  (GPL-3: imported to generate vectors, not vendored)
    W, K, D, N = 100, 20, 10, 100                               # onlinelda_test.py:40-43
    rs = np.random.RandomState(12345)
    # Hoffman's constructor keeps [a-z] only, so the vocabulary must be purely alphabetic
    vocab = ["w" + chr(97 + i // 26) + chr(97 + i % 26) for i in range(W)]
    hm = onlineldavb.OnlineLDA(vocab, K, D, 0.1, 0.3, 1024., 0.9)
    lam = np.asfortranarray(hm._lambda)
    docs = [[(int(w), int(rs.randint(10))) for w in rs.permutation(W)[:1 + rs.randint(N)]]
            for _ in range(D)]
    g0 = rs.gamma(100., 1. / 100., [K, D])
    docs0 = [tuple(zip(*doc)) for doc in docs]
    gh, sh = hm.do_e_step(docs0, max_steps=50, gamma=g0.T.copy())
    n = np.array([len(d) for d in docs])
    indptr = np.zeros(D + 1, np.int32)
    indptr[1:] = np.cumsum(n)
    flat = np.array([t for d in docs for t in d], dtype=np.int32)
    # the compiled reference on the same inputs (threshold 1e-3, 50 iterations)
    m = ref.online(W, K, D, alpha=.1, eta=.3)
    m.lambdas = lam
    gr, sr = m.estep(indptr, flat[:, 0].copy(), flat[:, 1].copy(), g0, 50, 1e-3)
    save("f7_hoffman_test_vi", K=K, V=W, B=D, indptr=indptr, ids=flat[:, 0].copy(),
         cnts=flat[:, 1].copy(), lam=lam, gamma0=g0, gamma_hoffman=np.asfortranarray(gh.T),
         sstats_hoffman=np.asfortranarray(sh), gamma_ref=gr, sstats_ref=sr)
    print("hoffman vs compiled reference: gamma %.2e sstats %.2e" % (
        np.max(np.abs(gh.T - gr) / np.abs(gr)),
        np.max(np.abs(sh - sr)[sr > 0] / sr[sr > 0])))

    # ---- F11: lower bound on the reference's own test_lower_bound set-up --------------
    # (onlinelda_test.py:72-95: K=22, W=100, D=30, 15 documents; the test asks for 1 %
    # agreement with Hoffman's approx_bound).  Values recorded:
    #   elbo_ref      the compiled reference (release build, -DNDEBUG as distutils passes:
    #                 lda.cpp:334 then reads psiLambda by row -- see DESIGN.md)
    #   elbo_oracle   oracle/cpu_ref.c with the column indexing (what the product computes);
    #                 the same restatement with reference_indexing=1 equals elbo_ref
    #   elbo_hoffman  onlineldavb.approx_bound on the reference's gamma
    W, K, D, N = 100, 22, 30, 60
    rs = np.random.RandomState(777)
    vocab = ["w" + chr(97 + i // 26) + chr(97 + i % 26) for i in range(W)]
    hm = onlineldavb.OnlineLDA(vocab, K, D, 0.1, 0.3, 1024., 0.9)
    lam = np.asfortranarray(hm._lambda)
    docs = [[(int(w), 1 + int(rs.randint(4))) for w in rs.permutation(W)[:1 + rs.randint(N)]]
            for _ in range(D // 2)]
    n = np.array([len(d) for d in docs])
    indptr = np.zeros(len(docs) + 1, np.int32)
    indptr[1:] = np.cumsum(n)
    flat = np.array([t for d in docs for t in d], dtype=np.int32)
    ids11, cnts11 = flat[:, 0].copy(), flat[:, 1].copy()
    m = ref.online(W, K, D, alpha=.1, eta=.3)
    m.lambdas = lam
    ref.seed(2024)
    elbo_ref = m.lower_bound(indptr, ids11, cnts11, -1, 100)
    ref.seed(2024)
    g11, s11 = m.estep(indptr, ids11, cnts11, None, 100)
    orc = pyoracle.Oracle()
    factor = D / float(len(docs))
    e_same = orc.lower_bound(lam, .1, .3, indptr, ids11, cnts11, g11, s11, factor, True)
    assert abs(e_same - elbo_ref) < 1e-10 * abs(elbo_ref), (e_same, elbo_ref)
    elbo_oracle = orc.lower_bound(lam, .1, .3, indptr, ids11, cnts11, g11, s11, factor, False)
    docs0 = [tuple(zip(*doc)) for doc in docs]
    elbo_hoffman = hm.approx_bound(docs0, gamma=np.ascontiguousarray(g11.T))
    # a second case with K > 64 and ids well beyond K, not from Hoffman
    K2, V2, B2 = 70, 900, 24
    ip2, ii2, cc2 = make_corpus(B2, V2, seed=99, mean_unique=40)
    ref.seed(5)
    m2 = ref.online(V2, K2, 1000, alpha=.1, eta=.3)
    lam2 = m2.lambdas
    ref.seed(6)
    elbo_ref2 = m2.lower_bound(ip2, ii2, cc2, 500, 100)
    ref.seed(6)
    g12, s12 = m2.estep(ip2, ii2, cc2, None, 100)
    e_same2 = orc.lower_bound(lam2, .1, .3, ip2, ii2, cc2, g12, s12, 500. / B2, True)
    assert abs(e_same2 - elbo_ref2) < 1e-10 * abs(elbo_ref2), (e_same2, elbo_ref2)
    elbo_oracle2 = orc.lower_bound(lam2, .1, .3, ip2, ii2, cc2, g12, s12, 500. / B2, False)
    save("f11_lower_bound", K=K, V=W, D=D, indptr=indptr, ids=ids11, cnts=cnts11, lam=lam,
         seed=2024, gamma_ref=g11, sstats_ref=s11, elbo_ref=elbo_ref, elbo_oracle=elbo_oracle,
         elbo_hoffman=elbo_hoffman,
         K2=K2, V2=V2, indptr2=ip2, ids2=ii2, cnts2=cc2, lam2=lam2, seed2=6, num_documents2=500,
         gamma_ref2=g12, sstats_ref2=s12, elbo_ref2=elbo_ref2, elbo_oracle2=elbo_oracle2)
    print("lower bound: reference %.6f  column-indexed %.6f  hoffman %.6f" % (
        elbo_ref, elbo_oracle, elbo_hoffman))


if __name__ == "__main__":
    main()
