"""Developer quick check on a GPU box: HIP E-step vs the CPU oracle + rough timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
from trlda_amd import _ffi
from trlda_amd.models import OnlineLDA
from trlda_amd.documents import CSRDocuments
from trlda_amd.utils.synthetic import make_corpus
from oracle.pyoracle import Oracle

def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))

o = Oracle()
L = _ffi.lib()
print("devices", _ffi.device_count())
cases = [(10, 1000, 100, 20, 1e-3, 30), (20, 300, 16, 50, 0.0, 60), (100, 7000, 200, 20, 1e-3, 100),
         (100, 7000, 200, 20, 0.0, 100), (7, 50, 5, 3, 1e-3, 10), (500, 2000, 12, 5, 1e-3, 100)]
for (K, V, B, it, thr, mu) in cases:
    indptr, ids, cnts = make_corpus(B, V, seed=3, mean_unique=mu)
    o.seed(5)
    lam = o.sample_gamma(K, V, 100) / 100.
    g0 = o.sample_gamma(K, B, 100) / 100.
    m = OnlineLDA(V, K, 1000)
    m.lambdas = lam
    docs = CSRDocuments(indptr, ids, cnts)
    g_o, s_o, it_o = o.estep(lam, 0.1, indptr, ids, cnts, g0, it, thr)
    for mode in (0, 1):
        L.trlda_model_set_sstats_mode(m._handle, mode)
        for T in (256, 64, 1024):
            L.trlda_model_set_doc_threads(m._handle, T)
            g, s, iters = m.update_variables(docs, latents=g0, max_iter=it, threshold=thr, return_iterations=True)
            nz = s_o > 0
            print("K=%d V=%d B=%d it=%d thr=%g mode=%d T=%d: gamma %.2e sstats %.2e zeros_ok %s iters_eq %s (mean %.1f)" % (
                K, V, B, it, thr, mode, T, relerr(g, g_o), relerr(s[nz], s_o[nz]), bool((s[~nz] == 0).all()),
                bool((iters == it_o).all()), iters.mean()))
    L.trlda_model_set_doc_threads(m._handle, 0)
    L.trlda_model_set_sstats_mode(m._handle, 0)

# timing at the bench shape
K, V, B = 100, 7000, 200
indptr, ids, cnts = make_corpus(B, V, seed=20150707, mean_unique=100)
o.seed(1)
lam = o.sample_gamma(K, V, 100) / 100.
g0 = o.sample_gamma(K, B, 100) / 100.
m = OnlineLDA(V, K, 1000000)
m.lambdas = lam
batch = m.upload(CSRDocuments(indptr, ids, cnts))
dev = m.device
def dalloc(nbytes):
    p = _ffi.vp(); _ffi.check(L.trlda_dev_alloc(dev, nbytes, C.byref(p))); return p
gam = dalloc(K * B * 8); sst = dalloc(K * V * 8)
g0f = np.asfortranarray(g0)
for mode in (0, 1):
    L.trlda_model_set_sstats_mode(m._handle, mode)
    for T in (64, 128, 256, 512, 1024):
        L.trlda_model_set_doc_threads(m._handle, T)
        for thr in (1e-3, 0.0):
            L.trlda_model_set_timing(m._handle, 0)
            def step():
                _ffi.check(L.trlda_dev_upload(dev, gam, g0f.ctypes.data, K * B * 8))
                _ffi.check(L.trlda_model_estep(m._handle, batch.handle, gam, sst, 20, thr, None))
            for _ in range(3): step()
            L.trlda_model_synchronize(m._handle)
            n = 50
            t = time.time()
            for _ in range(n): step()
            L.trlda_model_synchronize(m._handle)
            dt = (time.time() - t) / n
            L.trlda_model_set_timing(m._handle, 1)
            for _ in range(20): step()
            L.trlda_model_synchronize(m._handle)
            ks = []
            for w in range(4):
                us = C.c_double(); cnt = C.c_int64()
                L.trlda_model_get_timing(m._handle, w, C.byref(us), C.byref(cnt))
                ks.append(us.value / max(cnt.value, 1))
            print("mode=%d T=%4d thr=%g: %.1f us/step -> %.0f docs/s ; kernels us: rowsum %.1f eeb %.1f docs %.1f sstats %.1f" % (
                mode, T, thr, dt * 1e6, B / dt, *ks))
