"""Randomised check of whole update calls against the oracle composed step by step (GPU box; test
infrastructure): OnlineLDA.update_parameters over three calls per case -- random K, V, batch
sizes and document lengths (incl. empty documents, split documents, batches of one), trust-region
iterations 0 .. 4, inference iterations 1 .. 20, kappa / tau, with the fused launches on or off,
carried row sums on or off -- and BatchLDA epochs.

    python tests/fuzz_update.py [--cases 30] [--seed 1]        (tests/test_gpu_fuzz.py runs a short one)
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
    kind = rng.randint(4)
    lens = []
    for _ in range(B):
        r = rng.rand()
        if kind == 0:
            n = rng.randint(0, 140)
        elif kind == 1:
            n = int(np.exp(np.log(90) + .7 * rng.randn()))
        elif kind == 2:
            n = rng.randint(193, 700) if r < .15 else rng.randint(1, 128)
        else:
            n = rng.choice([0, 1, 128, 129, 144, 145, 192, 193, 257])
        lens.append(int(min(n, V)))
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = np.concatenate([rng.permutation(V)[:n] for n in lens] + [np.zeros(0, int)]).astype(np.int32)
    cnts = rng.randint(0 if rng.rand() < .3 else 1, 5, size=ip[-1]).astype(np.int32)   # zero counts are legal
    return CSRDocuments(ip, ids, cnts), lens


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--host-draw", action="store_true",
                    help="gamma0 from the host's glibc logarithms (bit for bit the oracle's) instead of the device "
                         "draw, whose logarithm may differ in the last bit: tells a flipped early exit from a defect")
    ap.add_argument("--per-call", type=int, default=-1, help="print lambda's error after every call of this case")
    args = ap.parse_args(argv)
    import trlda_amd
    from helpers import HipSampler, relerr
    from oracle.pyoracle import Oracle                 # the checker
    from test_gpu_update_loop import batch_model, online_model, oracle_online_update
    from trlda_amd import _ffi
    L = _ffi.lib()
    orc = Oracle()
    sampler = HipSampler(L)
    rng = np.random.RandomState(args.seed)
    worst = 0.0
    for case in range(args.cases):
        K = int(rng.choice([3, 20, 64, 100, 128, 129, 200, 333]))
        V = int(rng.choice([300, 2500, 9000, 40000]))
        eta = float(rng.choice([.01, .3]))
        alpha = float(rng.choice([.05, .5]))
        lam0 = np.asfortranarray(rng.gamma(100., .01, (K, V)))
        if case % 4 == 3:                                        # BatchLDA epochs
            B = int(rng.choice([1, 40, 300]))
            docs, lens = draw_docs(rng, B, V)
            epochs, inf = int(rng.choice([1, 3])), int(rng.choice([1, 10, 100]))
            m = batch_model(K, V, lam0, alpha=alpha, eta=eta)
            if args.host_draw:
                L.trlda_model_set_host_gamma_draw(m._handle, 1)
            trlda_amd.seed(900 + case)
            m.update_parameters(docs, max_epochs=epochs, max_iter_inference=inf)
            sampler.seed(900 + case)
            lam = lam0
            for _ in range(epochs):                              # batchlda.cpp:48-61
                g0 = sampler.sample_gamma(K, B, 100) / 100.
                _g, s, _it = orc.estep(lam, alpha, docs.indptr, docs.ids, docs.cnts, g0, inf, 1e-3, nthreads=8)
                lam = eta + s
            err = relerr(m.lambdas, lam)
            what = "BatchLDA B=%d epochs=%d inf=%d longest=%d" % (B, epochs, inf, max(lens))
        else:
            D = int(rng.choice([1000, 1000000]))
            kappa, tau = float(rng.choice([.6, .9])), float(rng.choice([10., 1024.]))
            fused, carry = int(rng.rand() < .7), int(rng.rand() < .7)
            m = online_model(K, V, lam0, D, alpha=alpha, eta=eta)
            if args.host_draw:
                L.trlda_model_set_host_gamma_draw(m._handle, 1)
            L.trlda_model_set_fused_update(m._handle, fused)
            L.trlda_model_set_carry_rowsums(m._handle, fused and carry)
            # (round 4: statistics inside the document launch never / for updates / always; the
            # longest lists cut into segments or not)
            merged, segs = int(rng.choice([0, 1, 2])), int(rng.rand() < .7)
            L.trlda_model_set_merged_launch(m._handle, merged)
            L.trlda_model_set_split_lists(m._handle, segs)
            lam = lam0
            shapes = []
            for call in range(3):
                B = int(rng.choice([1, 7, 64, 200, 224, 330]))
                docs, lens = draw_docs(rng, B, V)
                tr, inf = int(rng.choice([0, 1, 2, 4])), int(rng.choice([1, 5, 20]))
                shapes.append((B, tr, inf, max(lens)))
                trlda_amd.seed(100 * case + call)
                rho = m.update_parameters(docs, max_iter_tr=tr, max_iter_inference=inf, kappa=kappa, tau=tau)
                sampler.seed(100 * case + call)
                g0 = sampler.sample_gamma(K, B, 100) / 100.
                rho_o, lam, _g = oracle_online_update(orc, lam, alpha, eta, D, docs, g0, call, tr, inf,
                                                      kappa=kappa, tau=tau)
                if rho != rho_o:
                    print("MISMATCH case %d call %d: rho %r vs %r" % (case, call, rho, rho_o))
                    sys.exit(1)
                if call == 0:
                    # (the first call starts from the same lambda on both sides: rounding only)
                    first = relerr(m.lambdas, lam)
                    if not first < 1e-8:
                        print("MISMATCH case %d K=%d V=%d after the first call (B=%d tr=%d inf=%d): lambda %.2e"
                              % (case, K, V, B, tr, inf, first))
                        sys.exit(1)
                if case == args.per_call:
                    got = m.lambdas
                    d = np.abs(got - lam) / np.abs(lam)
                    k, w = np.unravel_index(np.argmax(d), d.shape)
                    act = np.zeros(V, bool); act[docs.ids] = True
                    print("   call %d (B=%d tr=%d inf=%d): lambda err %.2e at topic %d word %d (in this batch: %s; "
                          "lambda %.4g); over the batch's words %.2e, over the others %.2e" % (
                              call, B, tr, inf, d.max(), k, w, bool(act[w]), lam[k, w], d[:, act].max(),
                              d[:, ~act].max() if (~act).any() else 0.))
            err = relerr(m.lambdas, lam)
            what = "OnlineLDA fused=%d carry=%d merged=%d segments=%d (B, tr, inf, longest) %s" % (
                fused, carry, merged, segs, shapes)
        m.close()
        worst = max(worst, err)
        # (three calls in a row: each trust-region loop starts from the previous call's lambda and multiplies
        # what the two sides differ by -- 1e-10 after the first call, up to ~5e-8 after the third in the
        # worst of 1200 cases (seeds 503 / 504, the same with the host's bit-exact gamma0 and with round
        # 3's exp(psi)): the bound on the end of a case is 2e-7, the first call's stays 1e-8 above)
        if not err < 2e-7:
            print("MISMATCH case %d K=%d V=%d eta=%g alpha=%g %s: lambda %.2e" % (case, K, V, eta, alpha, what, err))
            sys.exit(1)
        print("case %3d ok (%.1e): K=%3d V=%5d %s" % (case, err, K, V, what), flush=True)
    print("all %d cases agree: worst lambda %.1e" % (args.cases, worst))
    return worst


if __name__ == "__main__":
    main()
