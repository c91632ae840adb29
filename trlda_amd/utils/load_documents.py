"""Text corpus loader: one document per line, ``"<n> id:cnt id:cnt ..."``.

Same file format, arguments and batching behaviour as the reference's
``trlda.utils.load_documents`` (python/utils/load_documents.py:6-69):

    6 5600:2 293:1 5548:1 2577:1 3733:3 2677:2

The file is parsed once, in C, by the library's host threads (``trlda_docs_from_text``,
include/trlda_hip.h) into CSR; ``load_documents`` then hands out the reference's lists of
``(word id, count)`` tuples, ``load_documents_csr`` the CSR batches themselves (what the HIP
path consumes, with no Python object per word).  A file the C parser does not accept
(anything but ``int:int`` tokens) goes through the reference's own Python steps, which raise
what the reference raises.
"""
import ctypes as C

import numpy as np

from .. import _ffi
from ..documents import CSRDocuments, csr_to_lists


def _parse_line(line):
    return [(int(wid), int(cnt)) for wid, cnt in
            (token.split(':') for token in line.split()[1:])]


def _parse_python(filepath):
    """The reference's per-line parsing (load_documents.py:38-47), to CSR."""
    lengths, ids, cnts = [], [], []
    with open(filepath) as handle:
        for line in handle:
            doc = _parse_line(line)
            lengths.append(len(doc))
            for wid, cnt in doc:
                ids.append(wid)
                cnts.append(cnt)
    offsets = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(np.asarray(lengths, dtype=np.int64), out=offsets[1:])
    return offsets, np.asarray(ids, dtype=np.int32), np.asarray(cnts, dtype=np.int32)


def parse_text(filepath):
    """Whole file -> ``(offsets[int64, num_docs + 1], ids[int32], cnts[int32])``."""
    L = _ffi.lib()
    handle = _ffi.vp()
    rc = L.trlda_docs_from_text(str(filepath).encode(), C.byref(handle))
    if rc != _ffi.OK:
        open(filepath).close()                       # a missing file raises here, as in the reference
        return _parse_python(filepath)               # raises on malformed tokens, like the reference
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


def _slice(parsed, lo, hi):
    offsets, ids, cnts = parsed
    p0, p1 = int(offsets[lo]), int(offsets[hi])
    return CSRDocuments((offsets[lo:hi + 1] - p0).astype(np.int32), ids[p0:p1], cnts[p0:p1])


def _batches(parsed, batch_size, stochastic, convert):
    """The control flow of the reference's generator (load_documents.py:31-62) over the parsed
    lines: the same batch boundaries, the same empty batches, the same ``poisson`` draws at
    the same points."""
    num_docs = len(parsed[0]) - 1
    current = int(np.random.poisson(batch_size)) if stochastic else batch_size
    start = 0
    for lineno in range(num_docs):
        if batch_size:
            while current == 0:
                yield convert(_slice(parsed, 0, 0))
                current = int(np.random.poisson(batch_size))
            if (lineno + 1) % current == 0:
                yield convert(_slice(parsed, start, lineno + 1))
                start = lineno + 1
                if stochastic:
                    current = int(np.random.poisson(batch_size))
    yield convert(_slice(parsed, start, num_docs))


def load_documents(filepath, batch_size=None, stochastic=False):
    """Load documents as lists of ``(word id, count)`` tuples.

    With ``batch_size`` a generator of batches is returned (the batch boundary rule
    ``(lineno + 1) % batch_size == 0`` and the trailing, possibly empty, batch are the
    reference's); ``stochastic=True`` draws each batch size from a Poisson
    distribution.  Without it, the whole file is returned as one list.
    """
    parsed = parse_text(filepath)
    if batch_size:
        return _batches(parsed, batch_size, stochastic, csr_to_lists)
    return next(_batches(parsed, batch_size, stochastic, csr_to_lists))


def load_documents_csr(filepath, batch_size=None, stochastic=False):
    """The same batches as :func:`load_documents`, as :class:`CSRDocuments` (int32 indptr /
    ids / counts): pass them straight to ``update_parameters`` / ``do_e_step``."""
    parsed = parse_text(filepath)
    if batch_size:
        return _batches(parsed, batch_size, stochastic, lambda csr: csr)
    return next(_batches(parsed, batch_size, stochastic, lambda csr: csr))
