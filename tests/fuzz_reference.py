"""Randomised check of whole update calls against THE REFERENCE ITSELF -- its unmodified C++ core
compiled into oracle/_ref/libtrlda_ref.so (oracle/Makefile; the library travels to the GPU box,
/root/reference does not and is not needed) -- through the same entry points a user calls:

  OnlineLDA.update_parameters     three calls per case; random max_iter_tr, max_iter_inference,
                                  kappa, tau, fixed or scheduled or adaptive rho, update_alpha,
                                  update_eta, init_gamma, min_alpha / min_eta
  BatchLDA.update_parameters      epochs with the alpha / eta line searches
  CumulativeLDA.update_parameters two calls
  do_e_step                       gamma and the statistics from a given gamma0 (K up to 200: the
                                  single-orientation kernel; documents of 300..600 words: split ones)

on random K, V, D, alpha (scalar or vector), eta, document lengths (empty documents, one-word
documents, documents beyond 128 / 192 words) and counts.  gamma0 comes from the seeded libc stream on
both sides (drawn on the device here: within 4e-15 of the host's values).  Compared: lambda, alpha,
eta, the returned rho, the update counter.

    python tests/fuzz_reference.py [--cases 30] [--seed 1]      (tests/test_gpu_fuzz.py runs a short one)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def draw_docs(rng, B, V):
    from trlda_amd.documents import CSRDocuments
    kind = rng.randint(3)
    lens = []
    for _ in range(B):
        r = rng.rand()
        if kind == 0:
            n = rng.randint(0, 90)
        elif kind == 1:
            n = rng.randint(129, 260) if r < .1 else rng.randint(1, 100)
        else:
            n = rng.choice([0, 1, 2, 64, 128, 129, 145, 193])
        lens.append(int(min(n, V)))
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = np.concatenate([rng.permutation(V)[:n] for n in lens] + [np.zeros(0, int)]).astype(np.int32)
    cnts = rng.randint(1, 6, size=ip[-1]).astype(np.int32)
    return CSRDocuments(ip, ids, cnts), lens


def rel(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    import trlda_amd
    from oracle.pyoracle import Reference               # the checker: the reference's own code
    from trlda_amd.models import BatchLDA, CumulativeLDA, OnlineLDA
    if not Reference.available():
        print("oracle/_ref/libtrlda_ref.so is not here (built where /root/reference exists)")
        return None
    ref = Reference()
    rng = np.random.RandomState(args.seed)
    worst = {"lambda": 0.0, "alpha": 0.0, "eta": 0.0}
    for case in range(args.cases):
        K = int(rng.choice([3, 10, 40, 100]))
        V = int(rng.choice([60, 400, 2500]))
        eta = float(rng.choice([.05, .3, 1.]))
        alpha = float(rng.choice([.05, .2])) if rng.rand() < .6 else rng.gamma(2., .1, K) + .01
        lam0 = np.asfortranarray(rng.gamma(100., .01, (K, V)))
        kind = ("online", "online", "online", "batch", "cumulative", "estep")[case % 6]
        what = ""
        if kind == "estep":
            K = int(rng.choice([7, 100, 129, 200]))
            V = int(rng.choice([700, 2500]))
            lam0 = np.asfortranarray(rng.gamma(100., .01, (K, V)))
            B = int(rng.choice([1, 30, 70]))
            docs, lens = draw_docs(rng, B, V)
            if rng.rand() < .6:                                   # a few long documents
                from trlda_amd.documents import CSRDocuments
                lens = list(lens)
                for d in rng.choice(B, size=min(B, 3), replace=False):
                    lens[d] = int(rng.randint(300, 600))
                ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
                ids = np.concatenate([rng.permutation(V)[:n] for n in lens]).astype(np.int32)
                docs = CSRDocuments(ip, ids, rng.randint(1, 6, size=ip[-1]).astype(np.int32))
            m = OnlineLDA(num_words=V, num_topics=K, num_documents=1000, alpha=alpha if np.isscalar(alpha) else .1,
                          eta=eta)
            r = ref.online(V, K, 1000, alpha=alpha if np.isscalar(alpha) else .1, eta=eta)
            m.lambdas = lam0
            r.lambdas = lam0
            g0 = np.asfortranarray(rng.gamma(100., .01, (K, B)))
            it, thr = int(rng.choice([0, 1, 20, 60])), float(rng.choice([1e-3, 1e-2]))
            g, st = m.do_e_step(docs, latents=g0, max_iter=it, threshold=thr)
            gr, sr = r.estep(docs.indptr, docs.ids, docs.cnts, gamma0=g0, max_iter=it, threshold=thr)
            big = sr > 1e-150
            errs = {"lambda": rel(st[big], sr[big]), "alpha": rel(g, gr), "eta": 0.0}
            what = " [B=%d longest=%d it=%d thr=%g]" % (B, max(lens), it, thr)
            m.close()
            for k, v in errs.items():
                worst[k] = max(worst[k], v)
            if not (errs["lambda"] < 1e-7 and errs["alpha"] < 1e-7):
                print("MISMATCH case %d estep K=%d V=%d: %s%s" % (case, K, V, errs, what))
                sys.exit(1)
            print("case %3d ok %-10s K=%3d V=%4d stats %.1e gamma %.1e%s" % (
                case, kind, K, V, errs["lambda"], errs["alpha"], what), flush=True)
            continue
        if kind == "online":
            D = int(rng.choice([500, 100000]))
            m = OnlineLDA(num_words=V, num_topics=K, num_documents=D, alpha=alpha, eta=eta)
            r = ref.online(V, K, D, alpha=alpha, eta=eta)
            m.lambdas = lam0
            r.lambdas = lam0
            for call in range(3):
                B = int(rng.choice([1, 5, 40, 120]))
                docs, lens = draw_docs(rng, B, V)
                kw = dict(max_iter_tr=int(rng.choice([0, 1, 3])), max_iter_inference=int(rng.choice([1, 5, 20])),
                          kappa=float(rng.choice([.6, .8])), tau=float(rng.choice([1., 64.])),
                          rho=float(rng.choice([-1., -1., .05])), adaptive=bool(rng.rand() < .3),
                          init_gamma=bool(rng.rand() < .8), update_alpha=bool(rng.rand() < .6),
                          update_eta=bool(rng.rand() < .6), min_alpha=float(rng.choice([1e-6, 1e-2])),
                          min_eta=float(rng.choice([1e-6, 1e-2])))
                s = 1000 * case + call
                trlda_amd.seed(s)
                rho = m.update_parameters(docs, **kw)
                ref.seed(s)
                rho_r = r.update_parameters(docs.indptr, docs.ids, docs.cnts, **kw)
                what += " [B=%d longest=%d tr=%d inf=%d a=%d e=%d ada=%d]" % (
                    B, max(lens), kw["max_iter_tr"], kw["max_iter_inference"], kw["update_alpha"],
                    kw["update_eta"], kw["adaptive"])
                if abs(rho - rho_r) > 1e-9 * abs(rho_r) or m.update_count != r.update_count:
                    print("MISMATCH case %d call %d: rho %r vs %r, count %d vs %d%s" % (
                        case, call, rho, rho_r, m.update_count, r.update_count, what))
                    sys.exit(1)
        elif kind == "batch":
            m = BatchLDA(num_words=V, num_topics=K, alpha=alpha, eta=eta)
            r = ref.batch(V, K, alpha=alpha, eta=eta)
            m.lambdas = lam0
            r.lambdas = lam0
            B = int(rng.choice([5, 60, 150]))
            docs, lens = draw_docs(rng, B, V)
            kw = dict(max_epochs=int(rng.choice([1, 3])), max_iter_inference=int(rng.choice([5, 30])),
                      max_iter_alpha=int(rng.choice([1, 5])), max_iter_eta=int(rng.choice([1, 8])),
                      update_alpha=bool(rng.rand() < .7), update_eta=bool(rng.rand() < .7),
                      min_alpha=1e-6, min_eta=1e-6)
            trlda_amd.seed(77 + case)
            m.update_parameters(docs, **kw)
            ref.seed(77 + case)
            r.update_parameters(docs.indptr, docs.ids, docs.cnts, **kw)
            what = " [B=%d longest=%d %s]" % (B, max(lens), kw)
        else:
            m = CumulativeLDA(num_words=V, num_topics=K, alpha=alpha, eta=eta)
            r = ref.cumulative(V, K, alpha=alpha, eta=eta)
            m.lambdas = lam0
            r.lambdas = lam0
            for call in range(2):
                B = int(rng.choice([5, 60]))
                docs, lens = draw_docs(rng, B, V)
                kw = dict(max_epochs=int(rng.choice([1, 2])), max_iter_inference=int(rng.choice([5, 30])),
                          max_iter_alpha=int(rng.choice([1, 5])), update_alpha=bool(rng.rand() < .7), min_alpha=1e-6)
                trlda_amd.seed(55 + 10 * case + call)
                m.update_parameters(docs, **kw)
                ref.seed(55 + 10 * case + call)
                r.update_parameters(docs.indptr, docs.ids, docs.cnts, **kw)
                what += " [B=%d longest=%d %s]" % (B, max(lens), kw)
        errs = {"lambda": rel(m.lambdas, r.lambdas), "alpha": rel(m.alpha, r.alpha), "eta": rel(m.eta, r.eta)}
        m.close()
        for k, v in errs.items():
            worst[k] = max(worst[k], v)
        if not (errs["lambda"] < 1e-7 and errs["alpha"] < 1e-7 and errs["eta"] < 1e-7):
            print("MISMATCH case %d %s K=%d V=%d: %s%s" % (case, kind, K, V, errs, what))
            sys.exit(1)
        print("case %3d ok %-10s K=%3d V=%4d lambda %.1e alpha %.1e eta %.1e%s" % (
            case, kind, K, V, errs["lambda"], errs["alpha"], errs["eta"], what[:150]), flush=True)
    # (do_e_step cases report the statistics under 'lambda' and gamma under 'alpha')
    print("all %d cases agree with the reference's own C++: worst %s" % (
        args.cases, {k: "%.1e" % v for k, v in worst.items()}))
    return max(worst.values())


if __name__ == "__main__":
    main()
