"""Diagnostic (GPU box): the timeline of ONE deferred launch (csrc/estep_merged.h): s_memrealtime
(100 MHz, one clock for the chip) at the start and end of every workgroup -- documents, the next
batch's preamble, the statistics of the step before -- and the CU each ran on.

    TRLDA_MERGED_STAMPS=1 python tools/deferred_stamps.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TRLDA_MERGED_STAMPS", "1")
from trlda_amd import _ffi                                            # noqa: E402
from trlda_amd.documents import CSRDocuments, DeviceBatch            # noqa: E402
from trlda_amd.utils.synthetic import make_corpus                     # noqa: E402

L = _ffi.lib()
L.trlda_debug_deferred_stamps.argtypes = [C.c_void_p, C.c_void_p]
K, V, B = 100, 7000, 200
L.trlda_seed(1)
lam = np.empty((K, V), order="F")
L.trlda_sample_gamma_init(K, V, lam)
model = _ffi.vp()
_ffi.check(L.trlda_model_create(C.byref(model), 0, K, V))
_ffi.check(L.trlda_model_set_lambda(model, lam))
_ffi.check(L.trlda_model_set_alpha(model, np.full(K, .1)))
_ffi.check(L.trlda_model_set_deferred_stats(model, 1))
csrs = [CSRDocuments(*make_corpus(B, V, seed=20150708 + i, mean_unique=100)) for i in range(8)]
csrs = [c for c in csrs if np.diff(c.indptr).max() <= 128][:4]        # register-kernel launches
batches = [DeviceBatch(c, V, 0) for c in csrs]
ptr = lambda n: (lambda p: (_ffi.check(L.trlda_dev_alloc(0, n, C.byref(p))), p)[1])(_ffi.vp())
g0 = np.empty((K, B), order="F")
L.trlda_sample_gamma_init(K, B, g0)
g0d, gd, sd = ptr(K * B * 8), ptr(K * B * 8), ptr(K * V * 8)
_ffi.check(L.trlda_dev_upload(0, g0d, g0.ctypes.data, g0.nbytes))
n = len(batches)
for i in range(40):
    _ffi.check(L.trlda_model_estep_io_next(model, batches[i % n].handle, batches[(i + 1) % n].handle, g0d, gd, sd,
                                           20, 0., None))
_ffi.check(L.trlda_model_synchronize(model))
buf = np.zeros(3 * 3072, dtype=np.uint64)
assert L.trlda_debug_deferred_stamps(model, buf.ctypes.data) == 0     # (clears the buffer)
_ffi.check(L.trlda_model_estep_io_next(model, batches[0].handle, batches[1].handle, g0d, gd, sd, 20, 0., None))
_ffi.check(L.trlda_model_estep_io_next(model, batches[1].handle, batches[2 % n].handle, g0d, gd, sd, 20, 0., None))
_ffi.check(L.trlda_model_synchronize(model))
assert L.trlda_debug_deferred_stamps(model, buf.ctypes.data) == 0
t = buf.reshape(3072, 3).astype(np.int64)
groups = {"statistics": t[:1024], "documents": t[1024:2048], "preamble": t[2048:]}
t0 = min(g[g[:, 0] > 0][:, 0].min() for g in groups.values() if (g[:, 0] > 0).any())
us = lambda x: (x - t0) / 100.0
print("last launch deferred flags %d (1 = left pending, 2 = carried)" % L.trlda_model_last_deferred(model))
for name, g in groups.items():
    g = g[g[:, 0] > 0]
    if not len(g):
        print("%-11s none" % name)
        continue
    dur = (g[:, 2] - g[:, 0]) / 100.0
    cus = len(set(int(v) >> 8 for v in g[:, 1]))          # XCC | SE | CU (drop wave / SIMD / pipe)
    print("%-11s %4d workgroups on %3d CUs: start %.2f..%.2f us (median %.2f), end %.2f..%.2f (median %.2f); "
          "a workgroup lasts %.2f..%.2f us (median %.2f, sum %.1f)" % (
              name, len(g), cus, us(g[:, 0].min()), us(g[:, 0].max()), us(np.median(g[:, 0])),
              us(g[:, 2].min()), us(g[:, 2].max()), us(np.median(g[:, 2])), dur.min(), dur.max(), np.median(dur),
              dur.sum()))
st = groups["statistics"]
st = st[st[:, 0] > 0]
if len(st):
    cnt = np.bincount(csrs[0].ids, minlength=V)
    N_long, N_short = int((cnt > 16).sum()), int(((cnt > 0) & (cnt <= 16)).sum())
    print("pending batch: %d short lists, %d long lists; statistics workgroups in order of index:" % (N_short, N_long))
    d = (st[:, 2] - st[:, 0]) / 100.0
    for lo in range(0, len(st), 16):
        print("  wg %3d..: start %s | lasts %s" % (lo, " ".join("%5.1f" % us(v) for v in st[lo:lo + 16, 0]),
                                                    " ".join("%4.1f" % v for v in d[lo:lo + 16])))
helper_cus = set(int(v) >> 8 for g in (groups["statistics"], groups["preamble"]) for v in g[g[:, 0] > 0][:, 1])
doc_cus = set(int(v) >> 8 for v in groups["documents"][groups["documents"][:, 0] > 0][:, 1])
print("CUs: documents %d, helpers %d, both %d" % (len(doc_cus), len(helper_cus), len(doc_cus & helper_cus)))

# where the helpers ran: per shader engine (XCC, SE) the documents it holds, the CUs its helpers used,
# the helper workgroups it was given and when the last of them ended
def se_of(v):
    v = int(v)
    return (v >> 16, (v >> 13) & 7)
docs_g = groups["documents"][groups["documents"][:, 0] > 0]
help_g = np.concatenate([g[g[:, 0] > 0] for g in (groups["statistics"], groups["preamble"])])
per = {}
for r in docs_g:
    per.setdefault(se_of(r[1]), [0, set(), 0, 0.0, 0.0])[0] += 1
for r in help_g:
    e = per.setdefault(se_of(r[1]), [0, set(), 0, 0.0, 0.0])
    e[1].add((int(r[1]) >> 8) & 15)
    e[2] += 1
    e[3] = max(e[3], us(r[2]))
    e[4] += (r[2] - r[0]) / 100.0
print("per shader engine: (xcc, se) documents | CUs its helpers ran on | helper workgroups | last helper ended (us) | helper CU time (us)")
for k in sorted(per):
    e = per[k]
    print("  %s docs %2d  helper CUs %d  helpers %3d  last end %5.1f  sum %6.1f" % (k, e[0], len(e[1]), e[2], e[3], e[4]))
