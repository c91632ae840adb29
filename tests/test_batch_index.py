"""The mini-batch index (csrc/batch_index.cpp: the flat form of the reference's Documents,
include/lda.h:21-23, plus the word-major order its statistics are added in, src/lda.cpp:207-213)
against golden digests taken from the builder of rounds 1-5 before it was rewritten (round 6):
every section of the buffer bitwise, every count the launch logic reads.  Host code: no GPU."""
import ctypes as C
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_index_golden",
                                              os.path.join(ROOT, "tests", "golden", "make_index_golden.py"))
gold = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gold)


@pytest.fixture(scope="module")
def lib(hip_lib):
    from trlda_amd import _ffi
    return C.CDLL(_ffi.LIB_PATH)


@pytest.fixture(scope="module")
def golden_index():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "f13_batch_index.json")))


def test_index_matches_the_golden_digests(lib, golden_index):
    cases = gold.cases()
    assert set(golden_index) == set(cases) | {"split_pays_32cus"}
    for name, (V, (ip, ii, cc)) in cases.items():
        got = gold.digest(lib, V, ip, ii, cc)
        want = golden_index[name]
        assert got["head"] == want["head"], name
        assert got["offsets"] == want["offsets"], name
        for sec in gold.SECTIONS:
            assert got["sections"][sec] == want["sections"][sec], (name, sec)
    V, (ip, ii, cc) = cases["split_pays"]
    assert gold.digest(lib, V, ip, ii, cc, cus=32) == golden_index["split_pays_32cus"]


def test_index_is_what_its_definition_says(lib):
    """... and against the definition itself on a small batch: a word's entries in document order,
    documents by decreasing length (stable), active words ascending, descriptors by decreasing
    length (stable in the word id), count sums."""
    from trlda_amd.utils.synthetic import make_corpus
    V = 300
    ip, ii, cc = make_corpus(50, V, seed=77, mean_unique=25)
    d = gold.digest(lib, V, ip, ii, cc)
    h = d["head"]
    info = np.zeros(64, np.int64)
    buf = np.zeros(h["total"], np.uint8)
    p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    assert lib.trlda_debug_batch_index(V, 50, p32(ip), p32(ii), p32(cc), 256,
                                       info.ctypes.data_as(C.POINTER(C.c_int64)), buf.ctypes.data, h["total"]) == 0
    off = dict(zip(gold.SECTIONS, d["offsets"]))
    i32 = lambda name, n: buf[off[name]:off[name] + 4 * n].view(np.int32)
    nnz = h["nnz"]
    doc_of = np.repeat(np.arange(50), np.diff(ip))
    order = np.lexsort((np.arange(nnz), ii))                   # by word, then by CSR position
    wrank = np.empty(nnz, np.int64); wrank[order] = np.arange(nnz)
    assert np.array_equal(i32("wrank", nnz), wrank)
    assert np.array_equal(i32("wdoc", nnz), doc_of[order])
    counts = np.bincount(ii, minlength=V)
    assert np.array_equal(i32("wptr", V + 1), np.r_[0, np.cumsum(counts)])
    assert np.array_equal(i32("active", h["n_active"]), np.nonzero(counts)[0])
    assert np.array_equal(buf[off["active_flag"]:off["active_flag"] + V], (counts > 0).astype(np.uint8))
    assert np.array_equal(i32("wc32", V), np.bincount(ii, weights=cc, minlength=V).astype(np.int64))
    lens = np.diff(ip)
    assert np.array_equal(i32("order", 50), np.argsort(-lens, kind="stable"))
    md = i32("mdesc", 4 * h["n_active"]).reshape(-1, 4)
    act = np.nonzero(counts)[0]
    short = act[counts[act] <= h["long_len"]]
    longs = act[counts[act] > h["long_len"]]
    want = np.r_[short[np.argsort(-counts[short], kind="stable")], longs[np.argsort(-counts[longs], kind="stable")]]
    assert np.array_equal(md[:, 0], want) and np.array_equal(md[:, 2], counts[want])
    assert np.array_equal(md[:, 1], np.r_[0, np.cumsum(counts)][want])


def test_index_rejects_what_the_reference_would_crash_on(lib):
    ip = np.array([0, 2], np.int32); cc = np.ones(2, np.int32)
    info = np.zeros(64, np.int64)
    pi = info.ctypes.data_as(C.POINTER(C.c_int64))
    p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    for bad in ([0, 5], [-1, 0]):
        ii = np.array(bad, np.int32)
        assert lib.trlda_debug_batch_index(5, 1, p32(ip), p32(ii), p32(cc), 256, pi, None, 0) != 0
    ii = np.array([0, 4], np.int32)
    assert lib.trlda_debug_batch_index(5, 1, p32(np.array([1, 2], np.int32)), p32(ii), p32(cc), 256, pi, None, 0) != 0
    assert lib.trlda_debug_batch_index(5, 1, p32(np.array([0, -1], np.int32)), p32(ii), p32(cc), 256, pi, None, 0) != 0
    assert lib.trlda_debug_batch_index(5, 1, p32(ip), p32(ii), p32(cc), 256, pi, None, 0) == 0
