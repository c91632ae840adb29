"""GPU box: host time to enqueue one E-step against its device time (python tools/enqueue_rate.py)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from trlda_amd import _ffi
from trlda_amd.documents import CSRDocuments, DeviceBatch
from trlda_amd.utils.synthetic import make_corpus
L = _ffi.lib()
K, V, B = 100, 7000, 200
L.trlda_seed(1)
lam = np.empty((K, V), order="F"); L.trlda_sample_gamma_init(K, V, lam)
model = _ffi.vp(); _ffi.check(L.trlda_model_create(C.byref(model), 0, K, V))
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
L.trlda_model_set_stream(model, _ffi.vp(stream))
L.trlda_model_set_lambda(model, lam); L.trlda_model_set_alpha(model, np.full(K, .1))
csr = CSRDocuments(*make_corpus(B, V, seed=5, mean_unique=100))
batch = DeviceBatch(csr, V, 0)
g0 = torch.rand(B * K, dtype=torch.float64, device=dev) + 0.5
g = torch.empty_like(g0); s = torch.empty(K * V, dtype=torch.float64, device=dev)
def step():
    _ffi.check(L.trlda_model_estep_io(model, batch.handle, g0.data_ptr(), g.data_ptr(), s.data_ptr(), 20, 1e-3, None))
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us/step, total %.1f us/step" % (1e6 * (t1 - t0) / 500, 1e6 * (t2 - t0) / 500))
