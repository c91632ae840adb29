"""Text corpus loader: one document per line, ``"<n> id:cnt id:cnt ..."``.

Same file format, arguments and batching behaviour as the reference's
``trlda.utils.load_documents`` (python/utils/load_documents.py:6-69):

    6 5600:2 293:1 5548:1 2577:1 3733:3 2677:2

``load_documents_csr`` parses straight to CSR (the form the HIP path consumes)
without building Python tuples.
"""
import numpy as np

from ..documents import CSRDocuments


def _parse_line(line):
    return [(int(wid), int(cnt)) for wid, cnt in
            (token.split(':') for token in line.split()[1:])]


def _batches(filepath, batch_size, stochastic):
    draw = (lambda: int(np.random.poisson(batch_size))) if stochastic else (lambda: batch_size)
    documents = []
    current = draw()
    with open(filepath) as handle:
        for lineno, line in enumerate(handle):
            documents.append(_parse_line(line))
            if batch_size:
                while current == 0:
                    # a Poisson draw of zero yields an empty batch (load_documents.py:50-52)
                    yield []
                    current = int(np.random.poisson(batch_size))
                if (lineno + 1) % current == 0:
                    yield documents
                    documents = []
                    if stochastic:
                        current = draw()
    yield documents


def load_documents(filepath, batch_size=None, stochastic=False):
    """Load documents as lists of ``(word id, count)`` tuples.

    With ``batch_size`` a generator of batches is returned (the batch boundary rule
    ``(lineno + 1) % batch_size == 0`` and the trailing, possibly empty, batch are the
    reference's); ``stochastic=True`` draws each batch size from a Poisson
    distribution.  Without it, the whole file is returned as one list.
    """
    if batch_size:
        return _batches(filepath, batch_size, stochastic)
    return next(_batches(filepath, batch_size, stochastic))


def load_documents_csr(filepath):
    """Whole file -> :class:`CSRDocuments` (int32 indptr / ids / counts)."""
    lengths, ids, cnts = [], [], []
    with open(filepath) as handle:
        for line in handle:
            tokens = line.split()[1:]
            lengths.append(len(tokens))
            for token in tokens:
                wid, cnt = token.split(':')
                ids.append(int(wid))
                cnts.append(int(cnt))
    indptr = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(np.asarray(lengths, dtype=np.int64), out=indptr[1:])
    return CSRDocuments(indptr, np.asarray(ids, dtype=np.int32), np.asarray(cnts, dtype=np.int32))
