"""The reference's own performance workload, code/trlda/python/tests/onlinelda_test.py:204-246
(test_speed): K = 100, W = 1000, 110 documents of 1..600 unique words with counts 0..9, initial
gamma ~ Gamma(100, 1/100), do_e_step(max_iter=100).  The reference asserts only that its C++ is
faster than Hoffman's NumPy code; here: the same call through the drop-in surface (list of tuples
in, NumPy out), the E-step alone on a resident batch, parity with the oracle (iteration counts
document by document), and the reference's C++ (oracle/_ref, when built) timed beside it.

    python tools/speed_workload.py          # on the GPU box, from the repo root
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import trlda_amd
    from oracle import pyoracle                     # the checker + the CPU baseline
    from trlda_amd import _ffi
    from trlda_amd.documents import as_csr
    from trlda_amd.models import OnlineLDA
    K, V, D, N = 100, 1000, 110, 600
    rng = np.random.RandomState(20150706)
    docs = []
    for _ in range(D):                              # onlinelda_test.py:228-233
        wordids = rng.permutation(V)[:1 + rng.randint(N)]
        docs.append([(int(w), int(rng.randint(10))) for w in wordids])
    g0 = np.asfortranarray(rng.gamma(100., 1. / 100., (K, D)))
    csr = as_csr(docs)
    lens = np.diff(csr.indptr)
    print("test_speed workload: K=%d W=%d, %d documents of %d..%d unique words (mean %.0f), "
          "max_iter=100" % (K, V, D, lens.min(), lens.max(), lens.mean()))
    trlda_amd.seed(1)
    model = OnlineLDA(num_words=V, num_topics=K, num_documents=10000, alpha=.1, eta=.3)
    lam = np.array(model.lambdas)
    L = _ffi.lib()

    def timed(fn, reps):
        fn()
        L.trlda_model_synchronize(model._handle)
        t = time.perf_counter()
        for _ in range(reps):
            out = fn()
        L.trlda_model_synchronize(model._handle)
        return (time.perf_counter() - t) / reps, out

    t_list, (g, s, it) = timed(lambda: model.do_e_step(docs, max_iter=100, latents=g0,
                                                       return_iterations=True), 20)
    batch = model.upload(csr)
    t_res, _ = timed(lambda: model.do_e_step(batch, max_iter=100, latents=g0), 20)
    print("GPU  do_e_step(list of tuples)    %8.3f ms  (%.0f documents/s)   [%s, fused preamble %d]"
          % (1e3 * t_list, D / t_list, L.trlda_model_last_doc_kernel(model._handle).decode(),
             L.trlda_model_last_preamble_fused(model._handle)))
    print("GPU  do_e_step(resident batch)    %8.3f ms  (%.0f documents/s)   (gamma0 up, gamma + "
          "sstats down)" % (1e3 * t_res, D / t_res))

    orc = pyoracle.Oracle()
    t = time.perf_counter()
    go, so, ito = orc.estep(lam, .1, csr.indptr, csr.ids, csr.cnts, g0, 100, 1e-3)
    t_orc = time.perf_counter() - t
    nz = so > 0
    print("parity vs oracle: gamma %.1e  sstats %.1e  iteration counts equal: %s  (mean %.1f "
          "iterations)" % (np.max(np.abs(g - go) / np.abs(go)),
                           np.max(np.abs(s[nz] - so[nz]) / so[nz]), np.array_equal(it, ito),
                           ito.mean()))
    print("CPU  oracle/cpu_ref.c, 1 thread   %8.3f ms  (%.0f documents/s)" % (1e3 * t_orc, D / t_orc))
    if pyoracle.Reference.available():
        rm = pyoracle.Reference().online(V, K, 10000, alpha=.1, eta=.3)
        rm.lambdas = lam
        rm.estep(csr.indptr, csr.ids, csr.cnts, g0, 100, 1e-3)
        t = time.perf_counter()
        for _ in range(3):
            rm.estep(csr.indptr, csr.ids, csr.cnts, g0, 100, 1e-3)
        t_ref = (time.perf_counter() - t) / 3
        print("CPU  reference C++ (oracle/_ref)  %8.3f ms  (%.0f documents/s)  -> GPU %.0fx "
              "(list in, arrays out)" % (1e3 * t_ref, D / t_ref, t_ref / t_list))


if __name__ == "__main__":
    main()
