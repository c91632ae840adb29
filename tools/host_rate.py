"""PCIe-inclusive rate of the host-pointer entry point (gamma0 up, gamma + sstats down per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trlda_amd import _ffi
from trlda_amd.models import OnlineLDA
from trlda_amd.documents import CSRDocuments
from trlda_amd.utils.synthetic import make_corpus
L = _ffi.lib()
K, V, B = 100, 7000, 200
docs = CSRDocuments(*make_corpus(B, V, seed=20150707, mean_unique=100))
L.trlda_seed(1)
m = OnlineLDA(V, K, 1000000)
g0 = np.empty((K, B), order="F"); L.trlda_sample_gamma_init(K, B, g0)
batch = m.upload(docs)
for _ in range(5): m.update_variables(batch, latents=g0, max_iter=20)
n = 200; t = time.perf_counter()
for _ in range(n): m.update_variables(batch, latents=g0, max_iter=20)
dt = (time.perf_counter() - t) / n
print("host-pointer do_e_step, device-resident batch: %.1f us/call -> %.0f docs/s" % (dt * 1e6, B / dt))
lst = docs.to_list()
t = time.perf_counter()
for _ in range(20): m.update_variables(lst, latents=g0, max_iter=20)
dt2 = (time.perf_counter() - t) / 20
print("host-pointer do_e_step, list-of-tuples docs:   %.1f us/call -> %.0f docs/s" % (dt2 * 1e6, B / dt2))
t = time.perf_counter()
for _ in range(20): m.update_parameters(batch, max_iter_tr=10, max_iter_inference=20)
dt3 = (time.perf_counter() - t) / 20
print("update_parameters(max_iter_tr=10):              %.1f us/call -> %.0f docs/s (includes the host-side libc-rand gamma draw)" % (dt3 * 1e6, B / dt3))
t = time.perf_counter()
for _ in range(20): m.update_parameters(batch, max_iter_tr=0, max_iter_inference=20)
dt4 = (time.perf_counter() - t) / 20
print("update_parameters(max_iter_tr=0):               %.1f us/call -> %.0f docs/s" % (dt4 * 1e6, B / dt4))
