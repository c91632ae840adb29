"""Documents: list-of-lists-of-(id, count) tuples  <->  CSR int32  <->  device batch.

Host-side counterpart of ``PyList_ToDocuments`` (reference
python/src/ldainterface.cpp:152-190): same accepted input, same ``TypeError``
messages, but the result is the flat CSR form of ``LDA::Documents``
(include/lda.h:21-23) that the HIP kernels consume.
"""
from collections.abc import MutableSequence
from itertools import chain

import numpy as np

from . import _ffi

try:                                                 # built by trlda_amd.build next to the .so
    from . import _fastdocs
except ImportError:                                  # host-side format conversion only: the
    _fastdocs = None                                 # Python path below does the same job


class CSRDocuments(object):
    """A batch already in CSR form: ``indptr[B+1]``, ``ids[nnz]``, ``cnts[nnz]`` (int32).

    Accepted wherever the reference accepts ``docs``; skips the per-tuple Python
    conversion (the fast path SURVEY.md §7 "host-side document conversion" asks for).
    """

    __slots__ = ("indptr", "ids", "cnts")

    def __init__(self, indptr, ids, cnts):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        self.ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.cnts = np.ascontiguousarray(cnts, dtype=np.int32)
        if self.indptr.ndim != 1 or len(self.indptr) < 1 or self.indptr[0] != 0:
            raise TypeError("indptr must be a 1-D array starting at 0.")
        if len(self.ids) != self.indptr[-1] or len(self.cnts) != self.indptr[-1]:
            raise TypeError("ids / cnts length must equal indptr[-1].")

    def __len__(self):
        return len(self.indptr) - 1

    def shard_cuts(self, world_size):
        """Document cut points ``c[0..world_size]`` of a contiguous split balanced by nnz
        (SURVEY.md §8e); rank r owns documents ``[c[r], c[r+1])``."""
        B = len(self)
        # equal shares of the cumulative entry count (+1 per document so that empty
        # documents still spread)
        weight = self.indptr.astype(np.int64) + np.arange(B + 1, dtype=np.int64)
        targets = weight[-1] * np.arange(world_size + 1, dtype=np.int64) // world_size
        cuts = np.searchsorted(weight, targets, side="left")
        cuts[0], cuts[-1] = 0, B
        return np.maximum.accumulate(cuts)

    def slice(self, lo, hi):
        p0, p1 = int(self.indptr[lo]), int(self.indptr[hi])
        return CSRDocuments(self.indptr[lo:hi + 1] - p0, self.ids[p0:p1], self.cnts[p0:p1])

    def shard(self, rank, world_size):
        """Contiguous range of documents for ``rank``."""
        if world_size <= 1:
            return self
        cuts = self.shard_cuts(world_size)
        return self.slice(int(cuts[rank]), int(cuts[rank + 1]))

    def to_list(self):
        return csr_to_lists(self)


class DocumentList(MutableSequence):
    """A batch in the reference's form -- a mutable sequence that indexes, iterates, compares,
    prints, sorts and shuffles like the list of lists of ``(word id, count)`` tuples that
    ``load_documents`` returns (python/utils/load_documents.py:31-69) -- that keeps its CSR arrays
    for as long as nobody has looked at a document: the tuples are built (all at once, in C) at the
    first look, and from then on the LISTS are the batch -- ``append`` / ``extend`` / ``sort`` /
    ``random.shuffle(batch)`` / ``batch[i].append((w, c))`` all count, and ``as_csr`` flattens the
    lists again instead of handing out arrays that no longer match them.  What the
    reference's README loop needs of a batch is ``update_parameters(documents)``: that path never
    builds a tuple.  It is not a ``list`` subclass (a list's storage cannot be filled lazily):
    ``isinstance(batch, list)`` is false and ``json.dumps`` wants ``list(batch)``."""

    __slots__ = ("_csr", "_lists")

    def __init__(self, csr):
        self._csr = csr
        self._lists = None

    @property
    def csr(self):
        """The batch as CSR arrays: the loader's own while no document has been handed out,
        re-flattened from the lists afterwards (a document may have been changed in place)."""
        if self._lists is not None:
            return as_csr(self._lists)
        return self._csr

    def to_list(self):
        if self._lists is None:
            self._lists = csr_to_lists(self._csr)
            self._csr = None
        return self._lists

    def __len__(self):
        return len(self._lists) if self._lists is not None else len(self._csr)

    def __getitem__(self, index):
        return self.to_list()[index]

    def __setitem__(self, index, value):
        self.to_list()[index] = value

    def __delitem__(self, index):
        del self.to_list()[index]

    def insert(self, index, value):
        self.to_list().insert(index, value)

    def sort(self, **kwargs):
        self.to_list().sort(**kwargs)

    def copy(self):
        return list(self.to_list())

    def __iter__(self):
        return iter(self.to_list())

    def __contains__(self, doc):
        return doc in self.to_list()

    def __eq__(self, other):
        if isinstance(other, DocumentList):
            other = other.to_list()
        return self.to_list() == other

    def __ne__(self, other):
        return not self == other

    __hash__ = None

    def __add__(self, other):
        return self.to_list() + list(other)

    def __radd__(self, other):
        return list(other) + self.to_list()

    def __mul__(self, n):
        return self.to_list() * n

    __rmul__ = __mul__

    def __repr__(self):
        return repr(self.to_list())

    def __reduce__(self):
        return (DocumentList, (self.csr,))


def csr_to_lists(csr):
    """CSR -> the reference's list of lists of ``(id, count)`` tuples."""
    if _fastdocs is not None:
        return _fastdocs.tuples(np.ascontiguousarray(csr.indptr), np.ascontiguousarray(csr.ids),
                                np.ascontiguousarray(csr.cnts))
    ip, ids, cnts = csr.indptr, csr.ids.tolist(), csr.cnts.tolist()
    return [list(zip(ids[ip[d]:ip[d + 1]], cnts[ip[d]:ip[d + 1]])) for d in range(len(csr))]


_INT_TYPES = {int, bool, np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32,
              np.uint64, np.intc, np.uintc, np.longlong, np.ulonglong}


def as_csr(docs):
    """Validate ``docs`` exactly as PyList_ToDocuments does and flatten it to CSR."""
    if isinstance(docs, CSRDocuments):
        return docs
    if isinstance(docs, (DeviceBatch, DocumentList)):
        return docs.csr
    if not isinstance(docs, list):
        raise TypeError("Documents must be stored in a list.")
    if _fastdocs is not None:
        # one pass in C (csrc/fastdocs.c); None when something is not a list / 2-tuple of ints,
        # which the Python path below then reports with PyList_ToDocuments' exceptions
        flat = _fastdocs.flatten(docs)
        if flat is not None:
            return CSRDocuments(np.frombuffer(flat[0], dtype=np.int32),
                                np.frombuffer(flat[1], dtype=np.int32),
                                np.frombuffer(flat[2], dtype=np.int32))
    lengths = np.empty(len(docs), dtype=np.int64)
    for i, doc in enumerate(docs):
        if not isinstance(doc, list):
            raise TypeError("Each document must be a list of tuples.")
        lengths[i] = len(doc)
    indptr = np.zeros(len(docs) + 1, dtype=np.int64)
    np.cumsum(lengths, out=indptr[1:])
    nnz = int(indptr[-1])
    if nnz >= 2 ** 31:
        raise TypeError("Not enough memory.")
    # Fast path: when every word is a 2-tuple of plain integers (checked with C-speed set
    # comprehensions) the whole batch is flattened by one np.fromiter; anything else goes
    # through the loop below, which raises what PyArg_ParseTuple would.
    words = chain.from_iterable(docs)
    if nnz and set(map(type, words)) == {tuple} and \
            set(map(len, chain.from_iterable(docs))) == {2} and \
            set(map(type, chain.from_iterable(chain.from_iterable(docs)))) <= _INT_TYPES:
        try:
            both = np.fromiter(chain.from_iterable(chain.from_iterable(docs)), dtype=np.int64,
                               count=2 * nnz).reshape(nnz, 2)
        except (OverflowError, ValueError):
            both = None
        if both is not None and np.abs(both).max() < 2 ** 31:
            return CSRDocuments(indptr.astype(np.int32), both[:, 0].astype(np.int32),
                                both[:, 1].astype(np.int32))
    flat = np.empty((nnz, 2), dtype=np.int32)
    pos = 0
    for doc in docs:
        for word in doc:
            # PyArg_ParseTuple(word, "ii", ...): a tuple of exactly two integers
            if not isinstance(word, tuple) or len(word) != 2:
                raise TypeError("Each document must be a list of tuples.")
            wid, cnt = word
            if isinstance(wid, (float, np.floating)) or isinstance(cnt, (float, np.floating)):
                raise TypeError("integer argument expected, got float")
            flat[pos, 0] = wid
            flat[pos, 1] = cnt
            pos += 1
    return CSRDocuments(indptr.astype(np.int32), flat[:, 0], flat[:, 1])


_BATCH_CREATE = None


def _batch_create():
    """trlda_batch_create taking plain addresses (a second ctypes object for the same symbol)"""
    global _BATCH_CREATE
    if _BATCH_CREATE is None:
        C = _ffi.C
        f = _ffi.lib()["trlda_batch_create"]
        f.restype = C.c_int
        f.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _BATCH_CREATE = f
    return _BATCH_CREATE


class DeviceBatch(object):
    """A batch resident in HBM (``trlda_batch``): CSR arrays + word-major index.

    Build once with ``model.upload(docs)`` and pass it as ``docs`` to
    ``update_parameters`` / ``update_variables`` to skip conversion and upload.
    """

    def __init__(self, docs, num_words, device):
        self.csr = csr = as_csr(docs)
        self.device = device
        self.num_words = num_words
        self.handle = _ffi.vp()
        # (CSRDocuments has made its arrays int32 and C-contiguous: their addresses go straight in --
        # numpy's ndpointer checks cost more per call than the library spends on this thread)
        _ffi.check(_batch_create()(_ffi.C.byref(self.handle), device, num_words, len(csr),
                                   csr.indptr.ctypes.data, csr.ids.ctypes.data, csr.cnts.ctypes.data))

    def __len__(self):
        return len(self.csr)

    @property
    def nnz(self):
        return int(self.csr.indptr[-1])

    def close(self):
        if getattr(self, "handle", None):
            _ffi.lib().trlda_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
