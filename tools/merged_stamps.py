"""Diagnostic (GPU box): where the time of a merged launch goes (csrc/estep_merged.h).  s_memrealtime
(100 MHz, one clock for the whole chip) of every document workgroup [start, end of its document, counted] and every statistics
workgroup [start, flag seen, end] of ONE launch at the headline shape.

    TRLDA_MERGED_STAMPS=1 python tools/merged_stamps.py [--update]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TRLDA_MERGED_STAMPS", "1")
from trlda_amd import _ffi                                            # noqa: E402
from trlda_amd.documents import CSRDocuments                          # noqa: E402
from trlda_amd.models import OnlineLDA                                # noqa: E402
from trlda_amd.utils.synthetic import make_corpus                     # noqa: E402

L = _ffi.lib()
L.trlda_debug_merged_stamps.argtypes = [C.c_void_p, C.c_void_p]
K, V, B = 100, 7000, 200
L.trlda_seed(1)
m = OnlineLDA(V, K, 1000000)
L.trlda_model_set_merged_launch(m._handle, 2)
docs = m.upload(CSRDocuments(*make_corpus(B, V, seed=20150707, mean_unique=100)))
g0 = np.empty((K, B), order="F")
L.trlda_sample_gamma_init(K, B, g0)
update = "--update" in sys.argv
for rep in range(4):
    if update:
        m.update_parameters(docs, max_iter_tr=3, max_iter_inference=20)
    else:
        m.update_variables(docs, latents=g0, max_iter=20, threshold=0.)
buf = np.zeros(3 * 1024, dtype=np.uint64)
assert L.trlda_debug_merged_stamps(m._handle, buf.ctypes.data) == 0
t = buf.reshape(1024, 3).astype(np.int64)
print("non-zero stamps: statistics rows %d, document rows %d; last kernel %s merged %d" % (
    int((t[:512, 0] > 0).sum()), int((t[512:, 0] > 0).sum()),
    L.trlda_model_last_doc_kernel(m._handle), L.trlda_model_last_merged(m._handle)))
d = t[512:512 + B]
h = t[:512]
h = h[h[:, 0] > 0]
t0 = d[:, 0].min()
us = lambda x: (x - t0) / 100.0
print("documents: start %.2f..%.2f us, end of document %.2f..%.2f (median %.2f), counted %.2f..%.2f"
      % (us(d[:, 0].min()), us(d[:, 0].max()), us(d[:, 1].min()), us(d[:, 1].max()),
         us(np.median(d[:, 1])), us(d[:, 2].min()), us(d[:, 2].max())))
print("statistics workgroups: %d; start %.2f..%.2f (median %.2f); flag seen %.2f..%.2f (median %.2f); "
      "end %.2f..%.2f (median %.2f)" % (len(h), us(h[:, 0].min()), us(h[:, 0].max()), us(np.median(h[:, 0])),
                                         us(h[:, 1].min()), us(h[:, 1].max()), us(np.median(h[:, 1])),
                                         us(h[:, 2].min()), us(h[:, 2].max()), us(np.median(h[:, 2]))))
early = h[h[:, 0] < d[:, 1].max()]
late = h[h[:, 0] >= d[:, 1].max()]
for name, g in (("resident before the last document ended", early), ("dispatched after it", late)):
    if len(g):
        print("  %-42s %3d: flag -> end %.2f us (median), start -> end %.2f" % (
            name, len(g), np.median(g[:, 2] - g[:, 1]) / 100., np.median(g[:, 2] - g[:, 0]) / 100.))
cnt = np.bincount(docs.csr.ids, minlength=V)
N_long = int((cnt > 16).sum()); N_short = int(((cnt > 0) & (cnt <= 16)).sum())
n_long = min(N_long, 128); n_short = max(1, min((N_short + 31) // 32, 256 - n_long))
hh = t[:n_short + n_long]
last_counted = d[:, 2].max()
for name, g in (("a wave per word (%d workgroups, %d words)" % (n_short, N_short), hh[:n_short]),
                ("a workgroup per long list (%d, %d words)" % (n_long, N_long), hh[n_short:])):
    print("  %-48s flag seen %.2f..%.2f after the last count; flag -> end median %.2f max %.2f; end %.2f..%.2f after the last count"
          % (name, (g[:, 1].min() - last_counted) / 100., (g[:, 1].max() - last_counted) / 100.,
             np.median(g[:, 2] - g[:, 1]) / 100., (g[:, 2] - g[:, 1]).max() / 100.,
             (g[:, 2].min() - last_counted) / 100., (g[:, 2].max() - last_counted) / 100.))
print("last document counted -> last statistics workgroup ended: %.2f us" % ((h[:, 2].max() - d[:, 2].max()) / 100.))
