"""GPU check of the other BASELINE.json configurations (reduced document counts so the CPU
oracle finishes in seconds): parity vs oracle + timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trlda_amd import _ffi
from trlda_amd.models import OnlineLDA, BatchLDA
from trlda_amd.documents import CSRDocuments
from trlda_amd.utils.synthetic import make_corpus
from oracle.pyoracle import Oracle
o = Oracle(); L = _ffi.lib()
def relerr(a, b): return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
for (name, K, V, B, it, mu) in [("config0 K=10 V=1000", 10, 1000, 100, 20, 50),
                                ("config3 K=200 V=50000", 200, 50000, 400, 100, 100),
                                ("config4 K=500 V=100000", 500, 100000, 256, 20, 100),
                                ("long docs K=100", 100, 7000, 64, 20, 400)]:
    ip, ii, cc = make_corpus(B, V, seed=11, mean_unique=mu)
    o.seed(3); lam = o.sample_gamma(K, V, 2) / 2.; g0 = o.sample_gamma(K, B, 100) / 100.
    t = time.time(); go, so, ito = o.estep(lam, .1, ip, ii, cc, g0, it, 1e-3, nthreads=8); tc = time.time() - t
    m = OnlineLDA(V, K, 100000); m.lambdas = lam
    batch = m.upload(CSRDocuments(ip, ii, cc))
    g, s, iters = m.update_variables(batch, latents=g0, max_iter=it, return_iterations=True)
    t = time.time()
    for _ in range(3): g, s, iters = m.update_variables(batch, latents=g0, max_iter=it, return_iterations=True)
    tg = (time.time() - t) / 3
    nz = so > 0
    print("%-26s B=%d max_n=%d: gamma %.2e sstats %.2e iters_eq %s mean_it %.1f | gpu(host-ptr) %.1f ms, cpu(8thr) %.0f ms" % (
        name, B, np.diff(ip).max(), relerr(g, go), relerr(s[nz], so[nz]), bool((iters == ito).all()), iters.mean(), tg * 1e3, tc * 1e3))
    m.close()
