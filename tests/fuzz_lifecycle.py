"""Randomised check of object LIFETIMES against the oracle (GPU box; test infrastructure): several
models alive at once -- each on its own stream, on torch's current stream or on a torch stream of
its own -- resident batches that outlive the model that last read them and are then read by another
model on another stream, models closed while their batches' device allocations sit in the upload
cache, E-steps and update calls interleaved between the models.  After every operation the result
is compared with the oracle's (which keeps its own copy of every model's lambda) and every host
array the library was handed is compared with a copy taken when it was made.

Why: round 4's long E-step fuzz found a caller's array changed under a later model (trlda_hip.hip,
batch_settle); that class of defect does not show in a fuzzer that uses one model at a time.

    python tests/fuzz_lifecycle.py [--steps 300] [--seed 1]     (tests/test_gpu_fuzz.py runs a short one)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Kept(object):
    """Host arrays handed to the library, each with the copy it must still equal."""

    def __init__(self):
        self.items = []

    def add(self, name, *arrays):
        for i, a in enumerate(arrays):
            self.items.append(("%s[%d]" % (name, i), a, a.copy()))

    def check(self, where):
        for name, a, c in self.items:
            if not np.array_equal(a, c):
                print("INPUT CHANGED after %s: %s differs at %s" % (where, name, np.nonzero(a.ravel() != c.ravel())[0][:8].tolist()))
                sys.exit(2)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args(argv)
    import torch
    import trlda_amd
    from fuzz_update import draw_docs
    from helpers import HipSampler, relerr
    from oracle.pyoracle import Oracle                 # the checker
    from test_gpu_update_loop import online_model, oracle_online_update
    from trlda_amd import _ffi
    from trlda_amd.documents import DeviceBatch
    L = _ffi.lib()
    orc = Oracle()
    sampler = HipSampler(L)
    rng = np.random.RandomState(args.seed)
    V = int(rng.choice([300, 2500]))                   # one vocabulary: any batch fits any model
    D = 100000
    kept = Kept()
    models, batches = [], []                           # dicts
    side_streams = [torch.cuda.Stream() for _ in range(2)]
    worst = 0.0
    counts = {}
    # host memory freed by the library or the runtime and taken again by the caller: after every
    # closed model a few thousand zeroed blocks of the sizes such objects have, watched for a while
    canaries = []                                      # (step they were made at, list of arrays)

    def check_canaries(where, now):
        for made, blocks in canaries:
            for a in blocks:
                if a.any():
                    print("STRAY WRITE after %s: a %d-byte block taken at step %d holds %s at %s" % (
                        where, a.nbytes, made, a[a != 0][:4].tolist(), np.nonzero(a)[0][:4].tolist()))
                    sys.exit(3)
        while canaries and now - canaries[0][0] > 12:
            canaries.pop(0)

    def new_model():
        K = int(rng.choice([7, 64, 100, 128, 200]))
        lam0 = np.asfortranarray(rng.gamma(100., .01, (K, V)))
        alpha, eta = float(rng.choice([.05, .5])), float(rng.choice([.01, .3]))
        m = online_model(K, V, lam0, D, alpha=alpha, eta=eta)
        where = int(rng.randint(3))                    # 0: a stream of the model's own
        if where == 1:
            L.trlda_model_set_stream(m._handle, _ffi.C.c_void_p(torch.cuda.current_stream().cuda_stream))
        elif where == 2:
            s = side_streams[int(rng.randint(len(side_streams)))]
            L.trlda_model_set_stream(m._handle, _ffi.C.c_void_p(s.cuda_stream))
        L.trlda_model_set_merged_launch(m._handle, int(rng.choice([0, 1, 2])))
        L.trlda_model_set_split_lists(m._handle, int(rng.rand() < .7))
        kept.add("lambda0", lam0)
        return {"m": m, "K": K, "lam": lam0, "alpha": alpha, "eta": eta, "calls": 0, "stream": where}

    def new_batch():
        B = int(rng.choice([1, 7, 64, 200, 224]))
        docs, lens = draw_docs(rng, B, V)
        kept.add("docs", docs.indptr, docs.ids, docs.cnts)
        return {"docs": docs, "dev": DeviceBatch(docs, V, 0) if rng.rand() < .8 else None, "B": B}

    for step in range(args.steps):
        ops = ["estep", "estep", "update"]
        if len(models) < 3:
            ops += ["new_model"] * (3 if not models else 1)
        if len(batches) < 5:
            ops += ["new_batch"] * (3 if not batches else 1)
        if models:
            ops.append("close_model")
        if batches:
            ops.append("close_batch")
        op = ops[int(rng.randint(len(ops)))]
        if op in ("estep", "update") and not (models and batches):
            op = "new_model" if not models else "new_batch"
        counts[op] = counts.get(op, 0) + 1
        note = ""
        if op == "new_model":
            models.append(new_model())
        elif op == "new_batch":
            batches.append(new_batch())
        elif op == "close_model":
            md = models.pop(int(rng.randint(len(models))))
            md["m"].close()
            canaries.append((step, [np.zeros(n // 4, np.int32) for n in range(32, 4097, 16) for _ in range(12)]))
        elif op == "close_batch":
            bt = batches.pop(int(rng.randint(len(batches))))
            if bt["dev"] is not None:
                bt["dev"].close()
        else:
            md = models[int(rng.randint(len(models)))]
            bt = batches[int(rng.randint(len(batches)))]
            m, K, docs = md["m"], md["K"], bt["docs"]
            arg = bt["dev"] if bt["dev"] is not None else docs
            if op == "estep":
                max_iter, thr = int(rng.choice([0, 1, 20])), float(rng.choice([0., 1e-3]))
                g0 = np.asfortranarray(rng.gamma(100., .01, (K, bt["B"])))
                g, s, it = m.update_variables(arg, latents=g0, max_iter=max_iter, threshold=thr, return_iterations=True)
                go, so, ito = orc.estep(md["lam"], md["alpha"], docs.indptr, docs.ids, docs.cnts, g0, max_iter, thr, nthreads=8)
                nz = so > 1e-150
                eg = float(np.max(np.abs(g - go) / np.abs(go))) if g.size else 0.
                es = float(np.max(np.abs(s[nz] - so[nz]) / so[nz])) if nz.any() else 0.
                err = max(eg, es / 10.)
                if not (eg < 1e-8 and es < 1e-7 and np.array_equal(it, ito)):
                    print("MISMATCH step %d: E-step of model K=%d (stream %d) on a batch of %d: gamma %.2e statistics %.2e "
                          "iterations equal %s" % (step, K, md["stream"], bt["B"], eg, es, np.array_equal(it, ito)))
                    sys.exit(1)
            else:
                tr, inf = int(rng.choice([0, 1, 3])), int(rng.choice([1, 5, 20]))
                seed = 7000 + 31 * step
                trlda_amd.seed(seed)
                rho = m.update_parameters(arg, max_iter_tr=tr, max_iter_inference=inf, kappa=.7, tau=64.)
                sampler.seed(seed)
                g0 = sampler.sample_gamma(K, bt["B"], 100) / 100.
                rho_o, md["lam"], _g = oracle_online_update(orc, md["lam"], md["alpha"], md["eta"], D, docs, g0,
                                                            md["calls"], tr, inf, kappa=.7, tau=64.)
                md["calls"] += 1
                err = relerr(m.lambdas, md["lam"])
                if rho != rho_o or not err < 1e-8:
                    print("MISMATCH step %d: update of model K=%d (stream %d, call %d) on a batch of %d, tr=%d inf=%d: "
                          "rho %r / %r, lambda %.2e" % (step, K, md["stream"], md["calls"], bt["B"], tr, inf, rho, rho_o, err))
                    sys.exit(1)
            worst = max(worst, err)
            note = " K=%d stream=%d B=%d resident=%d err=%.1e" % (K, md["stream"], bt["B"], bt["dev"] is not None, err)
        kept.check("step %d (%s)" % (step, op))
        check_canaries("step %d (%s)" % (step, op), step)
        if args.verbose:
            print("step %3d %-11s models=%d batches=%d%s" % (step, op, len(models), len(batches), note), flush=True)
    for md in models:
        md["m"].close()
    for bt in batches:
        if bt["dev"] is not None:
            bt["dev"].close()
    kept.check("the end")
    print("all %d steps agree (%s): worst %.1e" % (args.steps, ", ".join("%s %d" % kv for kv in sorted(counts.items())), worst))
    return worst


if __name__ == "__main__":
    main()
