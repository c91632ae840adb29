"""Randomised check of the E-step against the oracle (GPU box; test infrastructure): random K, V, batch
sizes, document lengths (empty documents, single words, every tier of the K <= 128 launch, split
documents, documents beyond the split range), duplicate ids, zero counts, iteration limits and
thresholds, both statistics modes, split documents on and off.

    python tests/fuzz_estep.py [--cases 60] [--seed 1]        (tests/test_gpu_fuzz.py runs a short one)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="run just this case (same random stream)")
    ap.add_argument("--keep-going", action="store_true")
    ap.add_argument("--run", default="", help="comma list of the cases to run (same random stream)")
    ap.add_argument("--two-kernel-preamble", type=int, default=0)
    ap.add_argument("--passes", default="", help="mode:split_docs:merged:split_lists,... instead of the four "
                                                 "standard passes over a case (diagnosis)")
    args = ap.parse_args(argv)
    from oracle.pyoracle import Oracle                 # the checker
    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.models import OnlineLDA
    L = _ffi.lib()
    orc = Oracle()
    rng = np.random.RandomState(args.seed)
    worst_g = worst_s = 0.0
    for case in range(args.cases):
        K = int(rng.choice([3, 7, 31, 64, 65, 100, 128, 129, 200, 257, 500]))
        V = int(rng.choice([50, 300, 2500, 9000]))
        B = int(rng.choice([1, 2, 17, 60, 300]))
        kind = rng.randint(5)
        lens = []
        for _ in range(B):
            r = rng.rand()
            if kind == 0:
                n = rng.randint(0, 150)
            elif kind == 1:
                n = rng.choice([0, 1, 127, 128, 129, 144, 145, 192, 193, 256, 257, 384, 385])
            elif kind == 2:
                n = rng.randint(150, 700) if r < .2 else rng.randint(1, 120)
            elif kind == 3:
                n = rng.choice([1024, 1025, 2048, 2049, 2500]) if r < .1 else rng.randint(0, 200)
            else:
                n = int(np.exp(np.log(100) + .6 * rng.randn()))
            lens.append(int(min(n, V if rng.rand() < .7 else 3 * V)))
        ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        ids = np.concatenate([rng.permutation(V)[:n] if n <= V else rng.randint(0, V, size=n)
                              for n in lens] + [np.zeros(0, int)]).astype(np.int32)
        cnts = rng.randint(0, 5, size=ip[-1]).astype(np.int32)
        lam = np.asfortranarray(rng.gamma(rng.choice([.3, 1., 100.]), rng.choice([1., .01]), (K, V)) + 1e-3)
        g0 = np.asfortranarray(rng.gamma(100., .01, (K, B)))
        max_iter = int(rng.choice([0, 1, 2, 20, 60]))
        thr = float(rng.choice([0., 1e-3, 1e-2]))
        alpha = float(rng.choice([.01, .1, 1.]))
        if args.only >= 0 and case != args.only:
            continue
        if args.run and case not in [int(x) for x in args.run.split(",")]:
            continue
        m = OnlineLDA.__new__(OnlineLDA)
        m._num_documents, m._update_count = 1000, 0
        m._ada_tau, m._ada_rho, m._ada_sq_norm = 1000., 1e-3, 1.
        m._setup(V, K, alpha, .3, None, _lambda=lam)
        kept = [x.copy() for x in (ip, ids, cnts, lam, g0)]
        go, so, ito = orc.estep(lam, alpha, ip, ids, cnts, g0, max_iter, thr, nthreads=8)
        docs = CSRDocuments(ip, ids, cnts)
        # (round 4: the statistics as workgroups of the document launch -- level 2 -- in every other
        # pass over a case; the longest lists cut into segments or not)
        L.trlda_model_set_split_preamble(m._handle, args.two_kernel_preamble)
        passes = [(mode, split, 2 if split else 0, int((case + split) % 2))
                  for mode in (0, 1) for split in (1, 0)]
        if args.passes:
            passes = [tuple(int(x) for x in t.split(":")) for t in args.passes.split(",")]
        for mode, split, merged, split_lists in passes:
            if True:
                L.trlda_model_set_sstats_mode(m._handle, mode)
                L.trlda_model_set_split_docs(m._handle, split)
                L.trlda_model_set_merged_launch(m._handle, merged)
                L.trlda_model_set_split_lists(m._handle, split_lists)
                g, s, it = m.update_variables(docs, latents=g0, max_iter=max_iter, threshold=thr,
                                              return_iterations=True)
                eg = float(np.max(np.abs(g - go) / np.abs(go))) if g.size else 0.0
                # (entries below 1e-150 come from exp(psi(lambda)) in the denormal range -- lambda
                # around 0.00135 -- where neither side has more than a few bits: bounded, not compared)
                nz = so > 1e-150
                es = float(np.max(np.abs(s[nz] - so[nz]) / so[nz])) if nz.any() else 0.0
                if args.passes:
                    print("   pass %s fused preamble %d" % ((mode, split, merged, split_lists),
                                                          L.trlda_model_last_preamble_fused(m._handle)))
                    import ctypes
                    from scipy.special import digamma as sp_psi
                    L.trlda_debug_peek.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
                    u = np.zeros((K, V), order="F")
                    L.trlda_debug_peek(m._handle, 0, u.ctypes.data, u.size)
                    epg = np.zeros((K, B), order="F")
                    L.trlda_debug_peek(m._handle, 1, epg.ctypes.data, epg.size)
                    tw = np.zeros(len(ids))
                    L.trlda_debug_peek(m._handle, 2, tw.ctypes.data, tw.size)
                    act = np.unique(ids)
                    uref = np.exp(sp_psi(lam[:, act]))
                    ud = u[:, act]
                    if not L.trlda_model_last_preamble_fused(m._handle):
                        uref = uref * np.exp(-sp_psi(lam.sum(axis=1)))[:, None]
                    big = uref > 1e-250
                    r = ud[big] / uref[big]
                    print("     u over the batch's words: ratio to scipy min %.6g max %.6g; entries at 2x: %d; "
                          "epg sum %.17g; tw sum %.17g" % (r.min(), r.max(), int((np.abs(r - 2) < 1e-6).sum()),
                                                         epg.sum(), tw.sum()))
                    if B == 1 and len(set(ids.tolist())) == len(ids):
                        order = np.argsort(ids, kind="stable")       # word order -> CSR position
                        tw_csr = np.zeros(len(ids)); tw_csr[order] = tw
                        eg0 = epg[:, 0]
                        ref = cnts / (u[:, ids].T @ eg0 + 1e-100)
                        off = np.nonzero(np.abs(tw_csr - ref) > 1e-9 * np.abs(ref))[0]
                        print("     tw: CSR slots off %s, ratio %s, counts there %s, next counts %s" % (
                            off.tolist(), [round(float(tw_csr[q] / ref[q]), 4) if ref[q] else -1 for q in off],
                            cnts[off].tolist(), cnts[np.minimum(off + 1, len(ids) - 1)].tolist()))
                    r2 = s[:, act][big] / np.maximum(so[:, act][big], 1e-300)
                    nzz = so[:, act][big] > 1e-250
                    print("     sstats: entries at 2x: %d of %d; at 1x: %d" % (
                        int((np.abs(r2[nzz] - 2) < 1e-6).sum()), int(nzz.sum()), int((np.abs(r2[nzz] - 1) < 1e-6).sum())))
                ok = eg < 1e-8 and es < 1e-7 and np.array_equal(it, ito) and \
                    bool((s[~nz] < 1e-149).all()) and bool(np.isfinite(s).all())
                worst_g, worst_s = max(worst_g, eg), max(worst_s, es)
                if not ok:
                    print("MISMATCH case %d K=%d V=%d B=%d kind=%d max_iter=%d thr=%g alpha=%g mode=%d "
                          "split=%d merged=%d split_lists=%d: gamma %.2e sstats %.2e iters_equal %s kernel %s split_wgs %d lens %s"
                          % (case, K, V, B, kind, max_iter, thr, alpha, mode, split, merged, split_lists, eg, es,
                             np.array_equal(it, ito), L.trlda_model_last_doc_kernel(m._handle).decode(),
                             L.trlda_model_last_split_workgroups(m._handle), sorted(lens)[-5:]))
                    bad = np.nonzero((np.abs(s - so) > 1e-7 * np.abs(so)).any(axis=0))[0]
                    rel = np.abs(s - so) / np.maximum(np.abs(so), 1e-300)
                    kk, ww = np.unravel_index(np.argsort(-rel, axis=None)[:5], rel.shape)
                    print("  worst entries (oracle, device, lambda): %s" % [
                        ("%.3e" % so[k, w], "%.3e" % s[k, w], "%.5g" % lam[k, w]) for k, w in zip(kk, ww)])
                    big = so > 1e-150
                    big[0, 0] = True
                    print("  largest relative error among entries above 1e-150: %.2e" % float(rel[big].max()))
                    owners = sorted({int(lens[d]) for d in range(B)
                                     if np.intersect1d(ids[ip[d]:ip[d + 1]], bad).size})
                    print("  words off: %d of %d; lengths of the documents holding them: %s; fused preamble %d"
                          % (bad.size, V, owners[:12] + ["..."] + owners[-6:],
                             L.trlda_model_last_preamble_fused(m._handle)
                             if hasattr(L, "trlda_model_last_preamble_fused") else -1))
                    if B == 1:
                        w = int(bad[0])
                        print("  word %d: ratio per topic %s" % (w, [round(float(s[k, w] / so[k, w]), 2) if so[k, w] > 0
                                                                      else -1 for k in range(K)]))
                        print("  word %d: lambda %s" % (w, ["%.4f" % lam[k, w] for k in range(K)]))
                        w = int(bad[-1])
                        print("  word %d: ratio per topic %s" % (w, [round(float(s[k, w] / so[k, w]), 2) if so[k, w] > 0
                                                                      else -1 for k in range(K)]))
                        print("  word %d: lambda %s" % (w, ["%.4f" % lam[k, w] for k in range(K)]))
                        pos = [int(np.nonzero(ids == w)[0][0]) for w in bad]
                        print("  positions in the document of the words off: %s; their counts %s; ratio %s" % (
                            sorted(pos), [int(cnts[q]) for q in sorted(pos)],
                            sorted({round(float(s[0, w] / so[0, w]), 3) for w in bad})))
                        print("  rank of those words among the document's ids: %s" % sorted(
                            int(np.searchsorted(np.sort(ids), w)) for w in bad))
                    if args.only >= 0 and mode == 0 and split == 1 and not args.passes:
                        for unfused in (0, 1):
                            L.trlda_model_set_split_preamble(m._handle, unfused)
                            offd = []
                            for d in range(B):
                                sub = CSRDocuments(np.array([0, lens[d]], np.int32), ids[ip[d]:ip[d + 1]],
                                                   cnts[ip[d]:ip[d + 1]])
                                g1, s1, _ = m.update_variables(sub, latents=np.asfortranarray(g0[:, d:d + 1]),
                                                               max_iter=max_iter, threshold=thr,
                                                               return_iterations=True)
                                _, s2, _ = orc.estep(lam, alpha, sub.indptr, sub.ids, sub.cnts,
                                                     np.asfortranarray(g0[:, d:d + 1]), max_iter, thr, nthreads=1)
                                nb = int(((np.abs(s1 - s2) > 1e-7 * np.abs(s2)).any(axis=0)).sum())
                                if nb:
                                    offd.append((lens[d], nb))
                            print("  one document at a time, two-kernel preamble %d: (length, words off) %s"
                                  % (unfused, offd))
                            g, s, it = m.update_variables(docs, latents=g0, max_iter=max_iter, threshold=thr,
                                                          return_iterations=True)
                            print("  whole batch, two-kernel preamble %d: words off %d" % (
                                unfused, int(((np.abs(s - so) > 1e-7 * np.abs(so)).any(axis=0)).sum())))
                        L.trlda_model_set_split_preamble(m._handle, 0)
                    if not args.keep_going:
                        sys.exit(1)
        m.close()
        # (the caller's arrays are inputs: round 4 found a document count changed under a later model --
        # the runtime writing through an event into a destroyed stream's memory, trlda_hip.hip batch_settle)
        for x, y, name in zip((ip, ids, cnts, lam, g0), kept, ("indptr", "ids", "cnts", "lambda", "gamma0")):
            if not np.array_equal(x, y):
                print("INPUT CHANGED case %d: %s differs at %s" % (case, name, np.nonzero(x != y)[0][:8].tolist()))
                sys.exit(2)
        print("case %3d ok: K=%3d V=%4d B=%3d kind=%d max_iter=%2d thr=%g longest=%4d" % (
            case, K, V, B, kind, max_iter, thr, max(lens) if lens else 0), flush=True)
    print("all %d cases agree: worst gamma %.1e, sstats %.1e" % (args.cases, worst_g, worst_s))
    return worst_g, worst_s


if __name__ == "__main__":
    main()
