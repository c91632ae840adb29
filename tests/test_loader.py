"""The text loader (reference python/utils/load_documents.py:6-69) against a fixture produced by
the reference's own loader (tests/golden/make_loader_golden.py): batch boundaries, the trailing
(possibly empty) batch, empty batches from zero Poisson draws, and where the global NumPy stream
is left.  CPU only: the parser is host code of libtrlda_hip.so (trlda_docs_from_text)."""
import os

import numpy as np
import pytest

from helpers import golden


@pytest.fixture(scope="module")
def fixture(tmp_path_factory, hip_lib):
    f = golden("f12_loader")
    path = tmp_path_factory.mktemp("corpus") / "data_train.dat"
    path.write_text(str(f["text"]))
    return f, str(path)


def flatten_lists(batches):
    sizes = [len(b) for b in batches]
    lens = [len(d) for b in batches for d in b]
    pairs = np.array([t for b in batches for d in b for t in d], dtype=np.int64).reshape(-1, 2)
    return np.array(sizes), np.array(lens), pairs[:, 0], pairs[:, 1]


def flatten_csr(batches):
    sizes = [len(b) for b in batches]
    lens = np.concatenate([np.diff(b.indptr) for b in batches]) if batches else np.zeros(0)
    ids = np.concatenate([b.ids for b in batches])
    cnts = np.concatenate([b.cnts for b in batches])
    return np.array(sizes), lens, ids, cnts


@pytest.mark.parametrize("form", ["lists", "csr"])
def test_loader_matches_the_reference_loader(fixture, form):
    from trlda_amd.utils import load_documents, load_documents_csr
    f, path = fixture
    for name in [str(c) for c in f["cases"]]:
        batch_size, stochastic, seed = [int(v) for v in f[name + "_args"]]
        np.random.seed(seed)
        loader = load_documents if form == "lists" else load_documents_csr
        got = loader(path, batch_size or None, bool(stochastic))
        if batch_size:
            assert not isinstance(got, list)         # a generator, like the reference's
            batches = list(got)
        else:
            batches = [got]
        sizes, lens, ids, cnts = (flatten_lists if form == "lists" else flatten_csr)(batches)
        assert np.array_equal(sizes, f[name + "_sizes"]), name
        assert np.array_equal(lens, f[name + "_lens"]), name
        assert np.array_equal(ids, f[name + "_ids"]) and np.array_equal(cnts, f[name + "_cnts"]), name
        assert np.random.randint(0, 2 ** 31 - 1) == int(f[name + "_next"]), name
        if form == "lists" and sizes.sum():
            doc = next(d for b in batches for d in b if d)
            assert isinstance(doc, list) and isinstance(doc[0], tuple) and type(doc[0][0]) is int


def test_parser_threads_and_line_endings(tmp_path, hip_lib, monkeypatch):
    """The file is cut at line ends and parsed by several host threads: same result for any
    thread count; CR LF line ends; a last line without a newline; an empty file."""
    from trlda_amd.utils.load_documents import _parse_python, parse_text
    rng = np.random.RandomState(0)
    lines = []
    for d in range(3000):
        n = rng.randint(0, 40)
        lines.append("%d %s" % (n, " ".join("%d:%d" % (rng.randint(0, 10 ** 6), rng.randint(0, 99))
                                              for _ in range(n))))
    for ending, tail in (("\n", "\n"), ("\r\n", "\r\n"), ("\n", "")):
        path = tmp_path / "c.dat"
        path.write_bytes((ending.join(lines) + tail).encode())
        want = _parse_python(path.read_bytes().decode())
        for threads in (1, 2, 7, 64):
            monkeypatch.setenv("TRLDA_PARSE_THREADS", str(threads))
            got = parse_text(str(path))
            for a, b in zip(got, want):
                assert np.array_equal(a, b), (ending, tail, threads)
    empty = tmp_path / "e.dat"
    empty.write_text("")
    off, ids, cnts = parse_text(str(empty))
    assert list(off) == [0] and len(ids) == 0 and len(cnts) == 0


def test_bounded_windows_lazy_batches_and_the_python_form(fixture, tmp_path, monkeypatch):
    """The file is read in bounded windows cut at line ends (a corpus larger than memory must
    stream, as it does in the reference): the same batches for any window size, nothing read
    before the first next(); a batch indexes, iterates and compares like the reference's list of
    lists of tuples while keeping its CSR arrays; without the built library the text still loads
    (the reference's own Python steps)."""
    from trlda_amd import _ffi
    from trlda_amd.documents import DocumentList, as_csr
    from trlda_amd.utils import load_documents, load_documents_csr
    f, path = fixture
    whole = load_documents(path)
    assert isinstance(whole, DocumentList) and len(whole) == len(list(whole))
    for chunk in (64, 1000, 1 << 16):
        got = list(load_documents(path, 7, chunk_bytes=chunk))
        want = list(load_documents(path, 7))
        assert len(got) == len(want) and all(a == b for a, b in zip(got, want)), chunk
        assert load_documents(path, chunk_bytes=chunk) == whole
        a, b = load_documents_csr(path, chunk_bytes=chunk), as_csr(whole)
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.ids, b.ids)
    # list behaviour of a batch
    batch = next(load_documents(path, 5))
    assert batch._lists is None and as_csr(batch) is batch.csr          # no tuple built so far
    first = batch[0]
    assert isinstance(first, list) and (not first or isinstance(first[0], tuple))
    assert batch == list(batch) and list(batch) == batch and not (batch != list(batch))
    assert batch[1:3] == list(batch)[1:3] and batch[-1] == list(batch)[-1]
    assert repr(batch) == repr(list(batch)) and batch + [[]] == list(batch) + [[]]
    # lazy: a missing file raises at the first next(), as the reference's generator does
    gen = load_documents(str(tmp_path / "missing.dat"), 5)
    with pytest.raises(IOError):
        next(gen)
    # no library: the Python form gives the same documents
    def missing():
        raise RuntimeError("libtrlda_hip.so not built")
    monkeypatch.setattr(_ffi, "lib", missing)
    assert load_documents(path, chunk_bytes=1000) == whole


def test_a_loaded_batch_is_a_mutable_sequence(fixture):
    """what load_documents returns must survive what callers do to the reference's
    plain lists -- shuffle, sort, append, extend, item assignment, del, changing a document in
    place -- and the models must then see the CHANGED batch, not the arrays the loader parsed."""
    import pickle
    import random
    from collections.abc import MutableSequence
    from trlda_amd.documents import as_csr
    from trlda_amd.utils import load_documents
    f, path = fixture
    batch = load_documents(path)
    plain = [list(d) for d in load_documents(path)]
    assert isinstance(batch, MutableSequence)
    untouched = as_csr(batch)
    assert batch._lists is None                      # still lazy: nothing was looked at
    random.seed(3)
    random.shuffle(batch)
    random.seed(3)
    random.shuffle(plain)
    assert batch == plain
    batch.append([(1, 2)])
    batch.extend([[(3, 4), (5, 6)], []])
    plain += [[(1, 2)], [(3, 4), (5, 6)], []]
    batch[0] = [(7, 1)]
    plain[0] = [(7, 1)]
    del batch[1]
    del plain[1]
    batch[2].append((9, 9))                          # a document changed in place
    plain[2].append((9, 9))
    batch.sort(key=len)
    plain.sort(key=len)
    batch.reverse()
    plain.reverse()
    assert batch == plain and len(batch) == len(plain)
    got, want = as_csr(batch), as_csr(plain)
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.ids, want.ids)
    assert np.array_equal(got.cnts, want.cnts)
    assert len(got) != len(untouched) or not np.array_equal(got.ids, untouched.ids)
    again = pickle.loads(pickle.dumps(batch))
    assert again == plain
    assert batch * 2 == plain * 2 and batch.copy() == plain


def test_malformed_corpus_raises_like_the_reference(tmp_path, hip_lib):
    from trlda_amd.utils import load_documents
    bad = tmp_path / "bad.dat"
    bad.write_text("2 1:1 2:2\n2 3:1 oops\n")
    with pytest.raises(ValueError):                  # `wid, wct = word.split(':')`
        load_documents(str(bad))
    bad.write_text("1 1:x\n")
    with pytest.raises(ValueError):                  # int('x')
        load_documents(str(bad))
    bad.write_text("1 1:2:3\n")
    with pytest.raises(ValueError):
        load_documents(str(bad))
    with pytest.raises(IOError):
        load_documents(str(tmp_path / "missing.dat"))
    # what Python's int() accepts and the C parser does not still loads (through the slow path)
    odd = tmp_path / "odd.dat"
    odd.write_text("1 1_0:+2\n")
    assert load_documents(str(odd)) == [[(10, 2)]]
    assert hip_lib.trlda_last_error()                # the C parser did refuse it


def test_list_flattening_extension(hip_lib):
    """csrc/fastdocs.c: the one-pass C flattening gives what the Python path gives, and whatever
    it does not take (floats, wrong shapes, big ints) still raises PyList_ToDocuments' errors
    (ldainterface.cpp:152-190)."""
    from trlda_amd import documents
    from trlda_amd.utils.synthetic import csr_to_docs, make_corpus
    assert documents._fastdocs is not None, "trlda_amd._fastdocs was not built"
    ip, ii, cc = make_corpus(50, 900, seed=4, mean_unique=20)
    docs = csr_to_docs(ip, ii, cc) + [[]]
    c = documents.as_csr(docs)
    assert np.array_equal(c.indptr, np.append(ip, ip[-1])) and np.array_equal(c.ids, ii)
    assert np.array_equal(c.cnts, cc)
    assert c.to_list() == docs
    assert len(documents.as_csr([])) == 0
    for bad, msg in (([[(1, 2.0)]], "integer argument expected"), ([[1]], "list of tuples"),
                     ([(1, 2)], "list of tuples"), ("x", "stored in a list"),
                     ([[(1, 2, 3)]], "list of tuples")):
        with pytest.raises(TypeError, match=msg):
            documents.as_csr(bad)
    assert documents.as_csr([[(np.int64(3), True)]]).ids[0] == 3      # NumPy ints: the slow path
