"""Golden digests of the mini-batch index (tests/golden/f13_batch_index.json).

Generated ONCE, in round 6, from the index builder as rounds 1-5 shipped it (the body of
trlda_batch_create, moved unchanged into csrc/batch_index.cpp), before that builder was rewritten for
speed: every section of the buffer (CSR copy, word-major order, padded id rows, split-document
layout, active words, flags, count sums, list descriptors, segment tasks) as a SHA-256 of its exact
bytes, plus the counts the launch logic reads.  tests/test_batch_index.py holds the builder to them.

    python tests/golden/make_index_golden.py        (no GPU: trlda_debug_batch_index is host code)
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

HEAD = ["V", "B", "nnz", "max_n", "n_active", "n_long", "long_len", "n_vl", "n_vl_tasks", "seg_len", "n_wg",
        "n_xrows", "max_list", "split_pays", "wc32_ok", "cnts_nonneg", "cls_short0", "cls_short1", "cls_short2",
        "cls_short3", "cls_long0", "cls_long1", "cls_long2", "cls_long3", "total"]
SECTIONS = ["indptr", "ids", "cnts", "order", "wrank", "wptr", "wdoc", "pad_meta", "pad_ids", "seg_meta",
            "seg_ids", "active", "long_words", "active_flag", "wc32", "mdesc", "vl_word", "vl_task", "vl_task_tiled"]


def section_bytes(h):
    B, V, nnz = h["B"], h["V"], h["nnz"]
    return [(B + 1) * 4, nnz * 4, nnz * 4, B * 4, nnz * 4, (V + 1) * 4, nnz * 4, B * 16, B * 144 * 4,
            h["n_wg"] * 32, h["n_wg"] * 144 * 4, h["n_active"] * 4, h["n_long"] * 4, V, V * 4,
            h["n_active"] * 16, h["n_vl"] * 16, h["n_vl_tasks"] * 16, h["n_vl_tasks"] * 16]


def cases():
    from trlda_amd.utils.synthetic import make_corpus
    rng = np.random.RandomState(6)

    def from_lengths(lengths, V, seed, cnt_hi=5, dup=False):
        r = np.random.RandomState(seed)
        ids, cnts, indptr = [], [], [0]
        for n in lengths:
            w = r.randint(0, V, size=n) if dup or n > V else r.choice(V, size=n, replace=False)
            ids.append(w)
            cnts.append(r.randint(0 if dup else 1, cnt_hi, size=n))
            indptr.append(indptr[-1] + n)
        cat = lambda a: np.concatenate(a).astype(np.int32) if a and sum(map(len, a)) else np.zeros(0, np.int32)
        return np.array(indptr, np.int32), cat(ids), cat(cnts)

    out = {}
    out["headline"] = (7000, make_corpus(200, 7000, seed=20150707, mean_unique=100))
    out["headline2"] = (7000, make_corpus(200, 7000, seed=3, mean_unique=100))
    out["config1"] = (1000, make_corpus(100, 1000, seed=11, mean_unique=60))
    out["empty_dup_zero"] = (30, from_lengths([0, 7, 0, 12, 3, 0], 30, 1, dup=True))
    out["one_word_vocab"] = (1, from_lengths([1, 0, 1, 1], 1, 2, dup=True))
    out["no_documents"] = (50, (np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)))
    out["tiers"] = (9000, from_lengths([64, 128, 129, 144, 145, 192, 193, 400, 2048, 2049, 3000, 100, 100], 9000, 4))
    out["split_pays"] = (9000, from_lengths([600, 500] + [80] * 198, 9000, 5))
    out["all_long"] = (9000, from_lengths([400] * 300, 9000, 7))
    out["batch1600"] = (7000, make_corpus(1600, 7000, seed=9, mean_unique=100))
    out["long_lists"] = (500, make_corpus(3000, 500, seed=10, mean_unique=60))
    out["very_long_lists"] = (300, from_lengths(list(rng.randint(20, 120, size=6000)), 300, 12, dup=True))
    ip, ii, cc = make_corpus(40, 400, seed=13, mean_unique=30)
    neg = cc.copy(); neg[::7] = -neg[::7]
    out["negative_counts"] = (400, (ip, ii, neg))
    big = cc.copy().astype(np.int64); big[:] = 2 ** 30
    out["count_sums_overflow"] = (400, (ip, ii, big.astype(np.int32)))
    out["dup_heavy"] = (64, from_lengths([300] * 20, 64, 14, dup=True))
    return out


def digest(lib, V, indptr, ids, cnts, cus=256):
    indptr = np.ascontiguousarray(indptr, np.int32); ids = np.ascontiguousarray(ids, np.int32)
    cnts = np.ascontiguousarray(cnts, np.int32)
    B = len(indptr) - 1
    info = np.zeros(64, np.int64)
    p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    f = lib.trlda_debug_batch_index
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int,
                  C.POINTER(C.c_int64), C.c_void_p, C.c_size_t]
    pi = info.ctypes.data_as(C.POINTER(C.c_int64))
    assert f(V, B, p32(indptr), p32(ids), p32(cnts), cus, pi, None, 0) == 0
    total = int(info[24])
    buf = np.full(total, 0xA5, np.uint8)
    assert f(V, B, p32(indptr), p32(ids), p32(cnts), cus, pi, buf.ctypes.data, total) == 0
    head = {k: int(info[i]) for i, k in enumerate(HEAD)}
    offs = [int(v) for v in info[32:52]]
    sec = {}
    for name, o, n in zip(SECTIONS, offs, section_bytes(head)):
        assert o + n <= total and o % 256 == 0
        sec[name] = hashlib.sha256(buf[o:o + n].tobytes()).hexdigest()[:24]
    return {"head": head, "offsets": offs, "sections": sec}


def main():
    from trlda_amd import _ffi
    lib = C.CDLL(_ffi.LIB_PATH)
    out = {}
    for name, (V, (ip, ii, cc)) in cases().items():
        out[name] = digest(lib, V, ip, ii, cc)
        if name == "split_pays":                       # (the one decision that depends on the chip's size)
            out[name + "_32cus"] = digest(lib, V, ip, ii, cc, cus=32)
    path = os.path.join(ROOT, "tests", "golden", "f13_batch_index.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        h = v["head"]
        print("%-22s B %5d nnz %7d active %5d long %4d/%3d vl %3d/%4d wg %3d split_pays %d total %8d" % (
            k, h["B"], h["nnz"], h["n_active"], h["n_long"], h["long_len"], h["n_vl"], h["n_vl_tasks"], h["n_wg"],
            h["split_pays"], h["total"]))


if __name__ == "__main__":
    main()
