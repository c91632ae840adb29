"""Host time of the mini-batch index alone (csrc/batch_index.cpp: plan + fill into a buffer, no
device): runs anywhere.

    python tools/index_rate.py [--batch 200] [--words 7000] [--reps 2000]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=200)
    ap.add_argument("--words", type=int, default=7000)
    ap.add_argument("--reps", type=int, default=2000)
    a = ap.parse_args()
    from trlda_amd import _ffi
    from trlda_amd.utils.synthetic import make_corpus
    lib = C.CDLL(_ffi.LIB_PATH)
    ip, ii, cc = make_corpus(a.batch, a.words, seed=20150707, mean_unique=100)
    f = lib.trlda_debug_batch_index_rate
    f.restype = C.c_double
    p32 = lambda x: x.ctypes.data_as(C.POINTER(C.c_int32))
    for mode, label in ((0, "plan"), (1, "plan + fill")):
        us = f(a.words, a.batch, p32(ip), p32(ii), p32(cc), 256, a.reps, mode)
        print("%-12s %7.2f us per batch of %d documents (%d entries, V = %d)" % (label, us, a.batch, len(ii), a.words))


if __name__ == "__main__":
    main()
