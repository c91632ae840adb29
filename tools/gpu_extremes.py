"""GPU box: E-step parity against the oracle on extreme inputs (lambda over 24 decades, tiny and
integer gamma, counts up to 1e6, document lengths at every tier boundary)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle.pyoracle import Oracle
from trlda_amd import _ffi
from trlda_amd.documents import CSRDocuments
from trlda_amd.models import OnlineLDA
L = _ffi.lib(); orc = Oracle()
rng = np.random.RandomState(0)
K, V, B = 100, 500, 12
ip=[0]; ids=[]; cnts=[]
for n in [5, 40, 100, 128, 130, 144, 150, 192, 200, 1, 0, 64]:
    ids += list(rng.permutation(V)[:n]); cnts += list(rng.randint(1, 1000000, size=n)); ip.append(ip[-1]+n)
ip,ids,cnts=np.array(ip,np.int32),np.array(ids,np.int32),np.array(cnts,np.int32)
cases = {
 "tiny lambda": np.asfortranarray(10.0**rng.uniform(-12, 2, (K, V))),
 "huge lambda": np.asfortranarray(10.0**rng.uniform(0, 12, (K, V))),
 "mixed": np.asfortranarray(10.0**rng.uniform(-8, 8, (K, V))),
 "integers": np.asfortranarray(rng.randint(1, 12, (K, V)).astype(float)),
}
bad = 0
for name, lam in cases.items():
    for gname, g0 in (("g~1", np.asfortranarray(rng.gamma(100, .01, (K, B)))), ("g tiny", np.asfortranarray(10.0**rng.uniform(-10, 0, (K, B)))), ("g ints", np.asfortranarray(rng.randint(1, 11, (K, B)).astype(float)))):
        m = OnlineLDA(num_words=V, num_topics=K, num_documents=100, alpha=.01, eta=.3); m.lambdas = lam
        g, s, it = m.update_variables(CSRDocuments(ip, ids, cnts), latents=g0, max_iter=30, threshold=1e-3, return_iterations=True)
        go, so, ito = orc.estep(lam, .01, ip, ids, cnts, g0, 30, 1e-3)
        fin = np.isfinite(go)
        eg = np.max(np.abs(g[fin] - go[fin]) / np.abs(go[fin])) if fin.any() else 0
        nz = so > 1e-300                      # below: denormals, which exp() on the device flushes
        es = np.max(np.abs(s[nz] - so[nz]) / so[nz]) if nz.any() else 0
        if es > 1e-7:
            badm = nz & (np.abs(s - so) > 1e-7 * so)
            print("   mismatching sstats entries: %d, oracle values there in [%.3g, %.3g], ours in [%.3g, %.3g]" % (
                badm.sum(), so[badm].min(), so[badm].max(), s[badm].min(), s[badm].max()))
        ok = eg < 1e-7 and es < 1e-7 and np.array_equal(np.isfinite(g), fin) and np.array_equal(it, ito)
        bad += not ok
        print("%-12s %-7s gamma %.1e sstats %.1e iters_eq %s finite_eq %s %s" % (name, gname, eg, es, np.array_equal(it, ito), np.array_equal(np.isfinite(g), fin), "ok" if ok else "MISMATCH"))
sys.exit(bad)
