import ctypes as C, os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from trlda_amd import _ffi
from trlda_amd.documents import CSRDocuments, DeviceBatch
from trlda_amd.utils.synthetic import SEED_BASE, make_corpus
L = _ffi.lib(); _ffi.require_gpu()
K, V, B, NB = 100, 7000, 200, 200
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
L.trlda_seed(1)
lam = np.empty((K, V), order="F"); L.trlda_sample_gamma_init(K, V, lam)
variant = sys.argv[1]
m = _ffi.vp(); _ffi.check(L.trlda_model_create(C.byref(m), 0, K, V))
if os.environ.get("OWN_STREAM"):                     # a stream of torch's pool instead of the null stream
    torch.cuda.set_stream(torch.cuda.Stream(device))
_ffi.check(L.trlda_model_set_stream(m, _ffi.vp(torch.cuda.current_stream(device).cuda_stream)))
_ffi.check(L.trlda_model_set_lambda(m, lam)); _ffi.check(L.trlda_model_set_alpha(m, np.full(K, .1)))
if variant == "benchlike":
    _ffi.check(L.trlda_model_set_sstats_mode(m, 0)); _ffi.check(L.trlda_model_set_doc_threads(m, 0))
    _ffi.check(L.trlda_model_set_dense_preamble(m, 0)); _ffi.check(L.trlda_model_set_split_preamble(m, 0))
    _ffi.check(L.trlda_model_set_word_sharding(m, 1))
csrs = [CSRDocuments(*make_corpus(B, V, seed=SEED_BASE + 1 + i, mean_unique=100)) for i in range(NB)]
batches = [DeviceBatch(c, V, 0) for c in csrs]
g0s = []
for i in range(NB):
    g0 = np.empty((K, B), order="F"); L.trlda_sample_gamma_init(K, B, g0)
    g0s.append(torch.from_numpy(np.ascontiguousarray(g0.T)).to(device))
_ffi.check(L.trlda_model_set_deferred_stats(m, 1)); _ffi.check(L.trlda_model_set_stream_lanes(m, 2))
outs = [(torch.empty(B * K, dtype=torch.float64, device=device), torch.empty(K * V, dtype=torch.float64, device=device)) for _ in range(2)]
up = (C.c_void_p * 2)()
thr = float(os.environ.get("THR", "0.001"))
def run(first, n):
    for i in range(first, first + n):
        j = i % NB
        up[0] = batches[(i + 1) % NB].handle.value; up[1] = batches[(i + 2) % NB].handle.value
        o = outs[i & 1]
        _ffi.check(L.trlda_model_estep_io_ahead(m, batches[j].handle, up, 2, g0s[j].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), 20, thr, None))
def fence():
    _ffi.check(L.trlda_model_flush(m)); torch.cuda.synchronize()
pos = 0
steps = int(os.environ.get("STEPS", "200"))
for _ in range(6):
    run(pos, steps); pos += steps; fence()
for rep in range(5):
    fence(); before = L.trlda_model_lane_steps(m); t0 = time.perf_counter()
    run(pos, 1); t1 = time.perf_counter()
    run(pos + 1, steps - 1); t2 = time.perf_counter()
    _ffi.check(L.trlda_model_flush(m)); t3 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0; pos += steps
    print("   host: first call %.1f us, the other %d calls %.1f us each, flush %.1f us, synchronize %.1f us"
          % ((t1 - t0) * 1e6, steps - 1, (t2 - t1) / (steps - 1) * 1e6, (t3 - t2) * 1e6, (dt - (t3 - t0)) * 1e6))
    print(variant, "thr", thr, "steps", steps, "%.2f us/step" % (dt / steps * 1e6), "through lanes", L.trlda_model_lane_steps(m) - before, flush=True)
