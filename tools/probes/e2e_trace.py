"""Host time of the end-to-end stream's calls one by one (GPU box): create (AHEAD ahead), E-step, destroy."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from trlda_amd import _ffi
from trlda_amd.documents import CSRDocuments, DeviceBatch
from trlda_amd.utils.synthetic import make_corpus
L = _ffi.lib()
K, V, B, N = 100, 7000, 200, 64
csrs = [CSRDocuments(*make_corpus(B, V, seed=100 + i, mean_unique=100)) for i in range(N)]
model = _ffi.vp()
_ffi.check(L.trlda_model_create(C.byref(model), 0, K, V))
lam = np.asfortranarray(np.random.RandomState(1).gamma(100., .01, (K, V)))
_ffi.check(L.trlda_model_set_lambda(model, lam))
_ffi.check(L.trlda_model_set_deferred_stats(model, 1))
_ffi.check(L.trlda_model_set_stream_lanes(model, 2))
dev = torch.device("cuda", 0)
g0 = torch.rand(B, K, dtype=torch.float64, device=dev) + .5
outs = [(torch.empty(B, K, dtype=torch.float64, device=dev), torch.empty(V, K, dtype=torch.float64, device=dev)) for _ in range(2)]
up = (C.c_void_p * 2)()
AHEAD = int(os.environ.get("AHEAD", "4"))
MODE = os.environ.get("MODE", "full")
pre = [DeviceBatch(c, V, 0) for c in csrs] if MODE == "nocreate" else None
win = {i: DeviceBatch(csrs[i % N], V, 0) for i in range(AHEAD)}
rows = []
if os.environ.get("NOGC"):
    import gc
    gc.collect(); gc.freeze(); gc.disable()
steps = int(os.environ.get("STEPS", "400"))
torch.cuda.synchronize()
T0 = time.perf_counter()
for i in range(steps):
    t0 = time.perf_counter()
    if MODE == "nocreate":
        win[i + AHEAD] = pre[(i + AHEAD) % N]
    else:
        win[i + AHEAD] = DeviceBatch(csrs[(i + AHEAD) % N], V, 0)
    t1 = time.perf_counter()
    up[0] = win[i + 1].handle.value; up[1] = win[i + 2].handle.value
    g, s = outs[i & 1]
    if MODE != "noestep":
        _ffi.check(L.trlda_model_estep_io_ahead(model, win[i].handle, up, 2, g0.data_ptr(), g.data_ptr(), s.data_ptr(), 20, 1e-3, None))
    elif i % 16 == 0:
        L.trlda_batch_long_word_len(win[i].handle)
    t2 = time.perf_counter()
    old = win.pop(i - 4, None)
    if old is not None and MODE != "nocreate":
        old.close()
    t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
_ffi.check(L.trlda_model_flush(model)); torch.cuda.synchronize()
T = time.perf_counter() - T0
r = np.array(rows[50:])
print(MODE, "AHEAD %d: %.1f us per step (%.2f M docs/s); host medians: create %.1f  estep %.1f  destroy %.1f; means %.1f %.1f %.1f; lane steps %d" % (
    AHEAD, T / steps * 1e6, B * steps / T / 1e6, *np.median(r, axis=0), *r.mean(axis=0), L.trlda_model_lane_steps(model)))
ul, us = C.c_double(), C.c_double()
L.trlda_model_lane_timing(model, C.byref(ul), C.byref(us))
print("   lane state %d, calibration: launch %.1f us, step %.1f us" % (L.trlda_model_lane_state(model), ul.value, us.value))
cnt = (C.c_longlong * 8)()
C.CDLL(_ffi.LIB_PATH).trlda_debug_ingest_counters(cnt)
print("   ingest counters [worker, taken over, cancelled, inline, hipMalloc, hipFree, stage held, stage upload]:", list(cnt))
ct = (C.c_double * 16)()
raw = C.CDLL(_ffi.LIB_PATH)
if hasattr(raw, "trlda_debug_call_times"):
    raw.trlda_debug_call_times(ct)
    n = max(ct[7], 1.0)
    print("   lane call, us per call: wait for the index %.1f, set-up %.1f, launch sequence %.1f (upload events %.1f, launch %.1f, reader marks %.1f), rest %.1f" % (
        ct[0] / n, ct[1] / n, ct[2] / n, ct[4] / n, ct[5] / n, ct[6] / n, ct[3] / n))
    nb_ = max(ct[15], 1.0)
    print("   a worker's build, us: queue %.1f, grace %.1f, plan + fill %.1f, allocation %.1f, wait for the allocation's last reader %.1f, copy call %.1f, publication %.1f (%d builds)" % (
        *[ct[8 + i] / nb_ for i in range(7)], int(ct[15])))
c = np.array(rows)[:, 0]
big = np.nonzero(c > 100)[0]
print("   creates > 100 us: %d of %d, at steps %s ..., sizes %s" % (len(big), len(c), list(big[:25]), [int(x) for x in c[big[:25]]]))
