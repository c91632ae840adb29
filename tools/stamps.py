"""Diagnostic: per-segment cycle shares of estep_docs_kernel (needs the -DTRLDA_STAMPS build).
   build + run: tools/stamps.sh (python -m trlda_amd.build --variant stamps -DTRLDA_STAMPS ...,
   selected through TRLDA_LIB)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trlda_amd import _ffi
assert "stamps" in os.path.basename(_ffi.LIB_PATH), "run through tools/stamps.sh (TRLDA_LIB=...stamps.so)"
from trlda_amd.models import OnlineLDA
from trlda_amd.documents import CSRDocuments
from trlda_amd.utils.synthetic import make_corpus
L = _ffi.lib()
L.trlda_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
K, V, B = (int(os.environ.get(k, d)) for k, d in (("STAMPS_K", "100"), ("STAMPS_V", "7000"), ("STAMPS_B", "200")))
WIDE = int(os.environ.get("STAMPS_WIDE", "0")) or K > 128 or int(os.environ.get("STAMPS_LEN", "0")) > 192
LEN = int(os.environ.get("STAMPS_LEN", "0"))         # every document exactly this long
ONE = int(os.environ.get("STAMPS_ONE", "0"))         # B - 1 documents of 100 words and one this long
lengths = np.full(B, LEN) if LEN else None
if ONE:
    lengths = np.full(B, 100)
    lengths[B // 2] = ONE
indptr, ids, cnts = make_corpus(B, V, seed=20150707, mean_unique=int(os.environ.get("STAMPS_MEAN", "95")),
                                lengths=lengths)
print("max doc length", np.diff(indptr).max())
L.trlda_seed(1)
m = OnlineLDA(V, K, 1000000)
g0 = np.empty((K, B), order="F"); L.trlda_sample_gamma_init(K, B, g0)
batch = m.upload(CSRDocuments(indptr, ids, cnts))
if WIDE:
    L.trlda_model_set_doc_kernel.argtypes = [C.c_void_p, C.c_int]
    L.trlda_model_set_doc_kernel(m._handle, 2)
names = ["stage beta", "E: dots + folds", "E/B: words past the unrolled groups", "B: axpy + part", "barrier 1", "psi", "barrier 2", "outputs"] if WIDE else ["psi: gnew + diffs + barrier", "stage beta", "product E (first)", "product B", "psi: barrier", "product E", "outputs", "psi: exp(psi) | reduce"]
for T in (0,):
    L.trlda_model_set_doc_threads(m._handle, T)
    m.update_variables(batch, latents=g0, max_iter=20, threshold=0.0)
    buf = np.zeros((B, 8), dtype=np.uint64)
    L.trlda_debug_read_stamps(buf.ctypes.data, B)
    m.update_variables(batch, latents=g0, max_iter=20, threshold=0.0)
    L.trlda_debug_read_stamps(buf.ctypes.data, B)
    tot = buf.astype(np.float64).sum(axis=1)
    print("kernel %s; slowest workgroup %.0f cycles, median %.0f" % (
        L.trlda_model_last_doc_kernel(m._handle).decode(), tot.max(), np.median(tot)))
    if ONE:
        # (workgroups are in order of decreasing length: the long document is workgroup 0)
        print("workgroup 0 (the %d-word document):" % ONE, " ".join("%.0f" % v for v in buf[0]))
        buf = buf[1:]
    mean = buf.astype(np.float64).mean(axis=0)
    print("T=%d  total %.0f cycles/doc" % (T, mean.sum()))
    for i, nme in enumerate(names[:8]):
        it_segs = (1, 2, 3, 4, 5, 6) if WIDE else (0, 3, 4, 5, 7)
        per = mean[i] / 20 if i in it_segs else mean[i]
        print("   %-38s %9.0f cycles (%4.1f%%)%s" % (nme, mean[i], 100 * mean[i] / mean.sum(),
              "  = %.0f / iteration" % per if i in it_segs else ""))
