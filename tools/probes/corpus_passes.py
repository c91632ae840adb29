"""trlda_model_estep_corpus pass after pass (GPU box): the time of each pass with what the library's section
clocks (TRLDA_CALL_TIMES=1) and ingest counters say about it -- where a slow pass loses its time."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("TRLDA_CALL_TIMES", "1")
import torch
from trlda_amd import _ffi
from trlda_amd.utils.synthetic import make_corpus
L = _ffi.lib()
raw = C.CDLL(_ffi.LIB_PATH)
K, V, B, N = 100, 7000, 200, int(os.environ.get("BATCHES", "200"))
parts = [make_corpus(B, V, seed=100 + i, mean_unique=100) for i in range(N)]
offs = np.zeros(N * B + 1, np.int64)
at = 0
for i, (ip, _, _) in enumerate(parts):
    offs[i * B:(i + 1) * B + 1] = ip.astype(np.int64) + at
    at += int(ip[-1])
ids = np.concatenate([p[1] for p in parts]); cnts = np.concatenate([p[2] for p in parts])
model = _ffi.vp()
_ffi.check(L.trlda_model_create(C.byref(model), 0, K, V))
lam = np.asfortranarray(np.random.RandomState(1).gamma(100., .01, (K, V)))
_ffi.check(L.trlda_model_set_lambda(model, lam))
dev = torch.device("cuda", 0)
g0 = (torch.rand(N * B, K, dtype=torch.float64, device=dev) * .2 + .9)
g = torch.empty_like(g0)
ring_t = [torch.empty(K * V, dtype=torch.float64, device=dev) for _ in range(4)]
ring = (C.c_void_p * 4)(*[t.data_ptr() for t in ring_t])
ct, ic = (C.c_double * 16)(), (C.c_longlong * 8)()


def counters():
    raw.trlda_debug_call_times(ct); raw.trlda_debug_ingest_counters(ic)
    return np.array(list(ct)), np.array(list(ic))


for rep in range(int(os.environ.get("PASSES", "12"))):
    torch.cuda.synchronize()
    c0, i0 = counters()
    t0 = time.perf_counter()
    _ffi.check(L.trlda_model_estep_corpus(model, N * B, offs.ctypes.data, ids.ctypes.data, cnts.ctypes.data, B,
                                          g0.data_ptr(), g.data_ptr(), ring, 4, 20, 1e-3, None))
    t1 = time.perf_counter()
    _ffi.check(L.trlda_model_synchronize(model)); torch.cuda.synchronize()
    t2 = time.perf_counter()
    c1, i1 = counters()
    d, di = c1 - c0, i1 - i0
    n = max(d[7], 1.0); nb = max(d[15], 1.0)
    print("pass %2d: %.1f us per step (calls returned after %.1f); lane call: wait for index %.1f, set-up %.1f, launch "
          "sequence %.1f (launch %.1f); builds: queue %.1f grace %.1f index %.1f; uploads: allocation %.1f wait for the allocation's last reader %.1f copy call %.1f; "
          "ingest [worker %d, taken over %d, hipMalloc %d, stage held %d, stage upload %d]; lane state %d" % (
              rep, (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6, d[0] / n, d[1] / n, d[2] / n, d[5] / n, d[8] / nb, d[9] / nb,
              d[10] / nb, d[11] / nb, d[12] / nb, d[13] / nb, di[0], di[1], di[4], di[6], di[7], L.trlda_model_lane_state(model)), flush=True)
