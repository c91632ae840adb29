"""Host time of the ingestion calls one by one (GPU box): where a loop of DeviceBatch(list).close()
spends its time with the index built on worker threads."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trlda_amd import _ffi
from trlda_amd.documents import CSRDocuments, DeviceBatch, as_csr
from trlda_amd.utils.synthetic import make_corpus
L = _ffi.lib()
V, B = 7000, 200
docs = CSRDocuments(*make_corpus(B, V, seed=20150707, mean_unique=100))
lst = docs.to_list()
for what in ("csr", "list"):
    for _ in range(20):
        DeviceBatch(lst if what == "list" else docs, V, 0).close()
    rows = []
    for i in range(40):
        t0 = time.perf_counter()
        c = as_csr(lst) if what == "list" else docs
        t1 = time.perf_counter()
        b = DeviceBatch(c, V, 0)
        t2 = time.perf_counter()
        b.close()
        t3 = time.perf_counter()
        rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
    r = np.array(rows)
    print(what, "median as_csr %.1f create %.1f close %.1f us; max %.1f %.1f %.1f" % (
        tuple(np.median(r, axis=0)) + tuple(r.max(axis=0))))
    print("   create:", " ".join("%.0f" % x for x in r[:, 1]))
    print("   close: ", " ".join("%.0f" % x for x in r[:, 2]))

# ---- do_e_step with list-of-tuples documents, call by call
from trlda_amd.models import OnlineLDA
L.trlda_seed(1)
K = 100
m = OnlineLDA(V, K, 1000000)
g0 = np.empty((K, B), order="F")
L.trlda_sample_gamma_init(K, B, g0)
for what in ("list", "csr"):
    arg = lst if what == "list" else docs
    for _ in range(5):
        m.update_variables(arg, latents=g0, max_iter=20)
    ts = []
    for i in range(30):
        t0 = time.perf_counter()
        m.update_variables(arg, latents=g0, max_iter=20)
        ts.append((time.perf_counter() - t0) * 1e6)
    print("do_e_step(%s): median %.0f us; all:" % (what, np.median(ts)), " ".join("%.0f" % x for x in ts))
