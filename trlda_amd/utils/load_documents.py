"""Text corpus loader: one document per line, ``"<n> id:cnt id:cnt ..."``.

Same file format, arguments and batching behaviour as the reference's
``trlda.utils.load_documents`` (python/utils/load_documents.py:6-69):

    6 5600:2 293:1 5548:1 2577:1 3733:3 2677:2

The file is read in bounded windows cut at line ends (``CHUNK_BYTES``; the reference streams
line by line, so a corpus larger than memory works there and must work here) and each window is
parsed in C by the library's host threads (``trlda_docs_from_buffer``, include/trlda_hip.h)
straight into CSR.  ``load_documents`` hands out batches that index like the reference's lists
of ``(word id, count)`` tuples but keep their CSR arrays (:class:`DocumentList`: tuples are built
only when a document is looked at, and ``update_parameters`` takes the arrays as they are);
``load_documents_csr`` the CSR batches themselves.  Text the C parser does not accept (anything
but ``int:int`` tokens), or a process without the built library, goes through the reference's
own Python steps, which raise what the reference raises.  Nothing is read before the first
``next()`` of the generator form, as in the reference.
"""
import ctypes as C
import io

import numpy as np

from .. import _ffi
from ..documents import CSRDocuments, DocumentList

CHUNK_BYTES = 64 << 20


def _parse_line(line):
    return [(int(wid), int(cnt)) for wid, cnt in
            (token.split(':') for token in line.split()[1:])]


def _parse_python(text):
    """The reference's per-line parsing (load_documents.py:38-47) of a piece of text, to CSR."""
    lengths, ids, cnts = [], [], []
    for line in io.StringIO(text, newline=None):     # universal newlines, like open() in text mode
        doc = _parse_line(line)
        lengths.append(len(doc))
        for wid, cnt in doc:
            ids.append(wid)
            cnts.append(cnt)
    offsets = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(np.asarray(lengths, dtype=np.int64), out=offsets[1:])
    return offsets, np.asarray(ids, dtype=np.int32), np.asarray(cnts, dtype=np.int32)


def _library():
    """The built library, or None: parsing text is host work and has a Python form."""
    try:
        return _ffi.lib()
    except RuntimeError:
        return None


def parse_bytes(data):
    """Whole lines of text -> ``(offsets[int64, num_docs + 1], ids[int32], cnts[int32])``."""
    L = _library()
    if L is not None:
        handle = _ffi.vp()
        if L.trlda_docs_from_buffer(data, len(data), C.byref(handle)) == _ffi.OK:
            try:
                n, nnz = L.trlda_docs_num_docs(handle), L.trlda_docs_nnz(handle)
                offsets = np.ctypeslib.as_array(L.trlda_docs_offsets(handle), shape=(n + 1,)).copy()
                if nnz:
                    ids = np.ctypeslib.as_array(L.trlda_docs_ids(handle), shape=(nnz,)).copy()
                    cnts = np.ctypeslib.as_array(L.trlda_docs_cnts(handle), shape=(nnz,)).copy()
                else:
                    ids, cnts = np.zeros(0, np.int32), np.zeros(0, np.int32)
            finally:
                L.trlda_docs_destroy(handle)
            return offsets, ids, cnts
    return _parse_python(data.decode())              # raises on malformed tokens, like the reference


def _pieces(filepath, chunk_bytes=None):
    """The file as consecutive parsed windows of at most ~chunk_bytes, cut after a line end."""
    chunk_bytes = chunk_bytes or CHUNK_BYTES
    with open(filepath, "rb") as handle:
        carry = b""
        while True:
            block = handle.read(chunk_bytes)
            if not block:
                break
            data = carry + block
            cut = data.rfind(b"\n") + 1
            if cut == 0:                             # a line longer than the window: keep reading
                carry = data
                continue
            carry = data[cut:]
            yield parse_bytes(data[:cut])
        if carry:
            yield parse_bytes(carry)


def parse_text(filepath):
    """Whole file -> ``(offsets[int64, num_docs + 1], ids[int32], cnts[int32])``."""
    parts = list(_pieces(filepath))
    if len(parts) == 1:
        return parts[0]
    if not parts:
        return np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32)
    offs = [parts[0][0]]
    for o, _, _ in parts[1:]:
        offs.append(o[1:] + offs[-1][-1])
    return np.concatenate(offs), np.concatenate([p[1] for p in parts]), \
        np.concatenate([p[2] for p in parts])


def _slice(parsed, lo, hi):
    offsets, ids, cnts = parsed
    p0, p1 = int(offsets[lo]), int(offsets[hi])
    return CSRDocuments((offsets[lo:hi + 1] - p0).astype(np.int32), ids[p0:p1], cnts[p0:p1])


_EMPTY = (np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32))


def _join(segments):
    """Documents [lo, hi) of several parsed windows as one CSR batch."""
    parts = [_slice(p, lo, hi) for p, lo, hi in segments if hi > lo]
    if not parts:
        return _slice(_EMPTY, 0, 0)
    if len(parts) == 1:
        return parts[0]
    indptr = [parts[0].indptr]
    for c in parts[1:]:
        indptr.append(c.indptr[1:] + indptr[-1][-1])
    return CSRDocuments(np.concatenate(indptr), np.concatenate([c.ids for c in parts]),
                        np.concatenate([c.cnts for c in parts]))


def _batches(pieces, batch_size, stochastic, convert):
    """The control flow of the reference's generator (load_documents.py:31-62) over the parsed
    windows: the same batch boundaries, the same empty batches, the same ``poisson`` draws at
    the same points."""
    current = int(np.random.poisson(batch_size)) if stochastic else batch_size
    open_batch = []                                  # (window, lo, hi) pieces of the batch being filled
    lineno = 0
    for parsed in pieces:
        num_docs = len(parsed[0]) - 1
        start = 0
        for i in range(num_docs):
            if batch_size:
                while current == 0:
                    yield convert(_join([]))
                    current = int(np.random.poisson(batch_size))
                if (lineno + 1) % current == 0:
                    yield convert(_join(open_batch + [(parsed, start, i + 1)]))
                    open_batch, start = [], i + 1
                    if stochastic:
                        current = int(np.random.poisson(batch_size))
            lineno += 1
        if start < num_docs:
            open_batch.append((parsed, start, num_docs))
    yield convert(_join(open_batch))


def load_documents(filepath, batch_size=None, stochastic=False, chunk_bytes=None):
    """Load documents as lists of ``(word id, count)`` tuples.

    With ``batch_size`` a generator of batches is returned (the batch boundary rule
    ``(lineno + 1) % batch_size == 0`` and the trailing, possibly empty, batch are the
    reference's); ``stochastic=True`` draws each batch size from a Poisson
    distribution.  Without it, the whole file is returned as one batch.  A batch is a
    :class:`DocumentList`: it indexes, iterates and compares like the reference's list of lists
    of tuples, and the models take its CSR arrays without converting anything.
    """
    if batch_size:
        return _batches(_pieces(filepath, chunk_bytes), batch_size, stochastic, DocumentList)
    return next(_batches(_pieces(filepath, chunk_bytes), batch_size, stochastic, DocumentList))


def load_documents_csr(filepath, batch_size=None, stochastic=False, chunk_bytes=None):
    """The same batches as :func:`load_documents`, as :class:`CSRDocuments` (int32 indptr /
    ids / counts): pass them straight to ``update_parameters`` / ``do_e_step``."""
    if batch_size:
        return _batches(_pieces(filepath, chunk_bytes), batch_size, stochastic, lambda csr: csr)
    return next(_batches(_pieces(filepath, chunk_bytes), batch_size, stochastic, lambda csr: csr))
