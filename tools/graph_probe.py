"""Experiment (GPU box): would a HIP graph shorten an update call?  One
OnlineLDA.update_parameters(max_iter_tr=10) call at the headline shape recorded once (stream
capture) and replayed, against the same call enqueued launch by launch.  DESIGN.md 7.

    python tools/graph_probe.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trlda_amd import _ffi                                            # noqa: E402
from trlda_amd.documents import CSRDocuments                          # noqa: E402
from trlda_amd.models import OnlineLDA                                # noqa: E402
from trlda_amd.utils.synthetic import make_corpus                     # noqa: E402

L = _ffi.lib()
L.trlda_debug_graph_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_double)]
for (K, V, B) in ((100, 7000, 200), (100, 7000, 1600), (500, 100000, 512)):
    for merged in (1, 0):
        for tr in (10, 0):
            L.trlda_seed(1)
            m = OnlineLDA(V, K, 1000000)
            L.trlda_model_set_merged_launch(m._handle, merged)
            docs = m.upload(CSRDocuments(*make_corpus(B, V, seed=20150707, mean_unique=100)))
            out = (C.c_double * 2)()
            rc = L.trlda_debug_graph_update(m._handle, docs.handle, 1000000, .3, tr, 20, 40, out)
            print("K=%d V=%d B=%d merged=%d max_iter_tr=%2d: %s" % (
                K, V, B, merged, tr, "direct %.1f us per call, graph replay %.1f us" % (out[0], out[1]) if rc == 0
                else "failed: " + L.trlda_last_error().decode()))
            docs.close()
            m.close()

# second form (round 5): a capture per call, hipGraphExecUpdate, launch -- two batches in turn
L.trlda_debug_graph_update2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int,
                                        C.c_int, C.POINTER(C.c_double)]
for (K, V, B) in ((100, 7000, 200),):
    for tr in (10, 0):
        L.trlda_seed(1)
        m = OnlineLDA(V, K, 1000000)
        d0 = m.upload(CSRDocuments(*make_corpus(B, V, seed=20150707, mean_unique=100)))
        d1 = m.upload(CSRDocuments(*make_corpus(B, V, seed=20150708, mean_unique=100)))
        out = (C.c_double * 4)()
        rc = L.trlda_debug_graph_update2(m._handle, d0.handle, d1.handle, 1000000, .3, tr, 20, 40, out)
        print("per-call capture + update: K=%d V=%d B=%d max_iter_tr=%2d: %s" % (
            K, V, B, tr, "direct %.1f us per call, graph %.1f us (host: %.1f us of capture + update per call, "
            "update refused in %.0f %% of the calls)" % (out[0], out[1], out[2], 100 * out[3]) if rc == 0
            else "failed: " + L.trlda_last_error().decode()))
        d0.close(); d1.close(); m.close()
