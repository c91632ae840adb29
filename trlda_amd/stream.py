"""A corpus pass of E-steps on a fixed lambda as ONE device-resident stream.

The reference runs such a pass as a Python loop over ``model.do_e_step(batch, latents=...)``
(python/src/ldainterface.cpp:311-390 -> LDA::updateVariablesVI, src/lda.cpp:160-220), host arrays in
and out of every call.  :class:`EStepStream` is the same sequence of calls for callers whose arrays
live on the device (torch tensors, or raw device addresses): the statistics of a call ride on the
next call's launch and two calls are in flight (``trlda_model_set_deferred_stats`` /
``trlda_model_set_stream_lanes`` / ``trlda_model_estep_io_ahead``, include/trlda_hip.h; DESIGN.md
3.7, 3.8) -- bitwise the results of the loop, at 26 instead of ~240 microseconds per 200-document
batch.  Results are complete after :meth:`flush` (or when the ``with`` block ends).

    with EStepStream(model) as s:
        for i, b in enumerate(batches):                    # DeviceBatch objects (model.upload)
            s.step(b, batches[i + 1:i + 3], gamma0[i], gamma[i], sstats[i])
    # gamma[i] (K x B column-major: a document's K values contiguous) and sstats[i] (K x V column-major:
    # a word's K values contiguous) are complete here.  As row-major torch tensors these are shapes
    # (B, K) and (V, K).

Consecutive calls must write different arrays (two are in flight); a caller that reuses one set
of arrays is recognised by the library and goes one call at a time.
"""
import ctypes as C

from . import _ffi

__all__ = ["EStepStream", "corpus_pass"]


def _address(x, what="array", numel=None, dtype="float64", device=None):
    """device address of a torch tensor (or anything with data_ptr()), or an int / None.

    A tensor is checked before its address is handed to the kernels -- on the device (the model's),
    of the element type the kernels read (`dtype`), contiguous, of exactly `numel` elements: a host
    tensor or a float32 one would otherwise be a GPU fault or silent garbage, not an error.  A raw
    address (int) is the caller's word."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        if hasattr(x, "is_contiguous") and not x.is_contiguous():
            raise TypeError("%s must be contiguous." % what)
        if hasattr(x, "is_cuda") and not x.is_cuda:
            raise TypeError("%s must live on the device (got a host tensor)." % what)
        got = str(getattr(x, "dtype", dtype)).replace("torch.", "")
        if got != dtype:
            raise TypeError("%s must be %s (got %s)." % (what, dtype, got))
        if numel is not None and hasattr(x, "numel") and int(x.numel()) != int(numel):
            raise ValueError("%s must have %d elements (got %d)." % (what, numel, int(x.numel())))
        dev = getattr(getattr(x, "device", None), "index", None)
        if device is not None and dev is not None and int(dev) != int(device):
            raise ValueError("%s lives on device %d, the model on device %d." % (what, dev, device))
        return x.data_ptr()
    return int(x)


class EStepStream(object):
    def __init__(self, model, lanes=2, deferred=True):
        self._model = model
        self._lib = _ffi.lib()
        self._up = (C.c_void_p * 2)()
        self._steps = 0
        _ffi.check(self._lib.trlda_model_set_deferred_stats(model._handle, int(bool(deferred))))
        _ffi.check(self._lib.trlda_model_set_stream_lanes(model._handle, int(lanes)))
        self._open = True

    def step(self, batch, upcoming, gamma0, gamma, sstats, max_iter=100, threshold=1e-3, iterations=None):
        """One LDA::updateVariablesVI call (src/lda.cpp:160-220) on `batch` (a DeviceBatch).

        upcoming     the batches of the next calls, in order (at most two are looked at; may be empty)
        gamma0       K x B initial latents on the device (a document's K values contiguous), read only
        gamma        K x B result; sstats: K x V result (a word's K values contiguous); both fp64
        iterations   optional int32[B] on the device: executed iterations per document
        """
        if not self._open:
            raise RuntimeError("the stream is closed.")
        n = 0
        for b in list(upcoming)[:2]:
            if b is None:
                break
            self._up[n] = b.handle.value
            n += 1
        K, V, B = self._model.num_topics, self._model.num_words, len(batch)
        dev = getattr(self._model, "_device", None)
        _ffi.check(self._lib.trlda_model_estep_io_ahead(
            self._model._handle, batch.handle, self._up, n,
            _address(gamma0, "gamma0", K * B, device=dev), _address(gamma, "gamma", K * B, device=dev),
            _address(sstats, "sstats", K * V, device=dev), int(max_iter), float(threshold),
            _address(iterations, "iterations", B, dtype="int32", device=dev)))
        self._steps += 1

    def flush(self):
        """everything so far is complete on the model's stream (enqueued: no host wait)"""
        _ffi.check(self._lib.trlda_model_flush(self._model._handle))

    def synchronize(self):
        _ffi.check(self._lib.trlda_model_synchronize(self._model._handle))

    @property
    def steps_through_lanes(self):
        return int(self._lib.trlda_model_lane_steps(self._model._handle))

    def close(self):
        if self._open:
            self._open = False
            self.flush()
            _ffi.check(self._lib.trlda_model_set_stream_lanes(self._model._handle, 1))
            _ffi.check(self._lib.trlda_model_set_deferred_stats(self._model._handle, 0))

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
            return False
        try:                                             # (an exception is on its way out of the block:
            self.close()                                 # a failing flush must not take its place)
        except Exception:
            pass
        return False


def corpus_pass(model, offsets, ids, cnts, batch_size, gamma0, gamma, sstats_ring, max_iter=100,
                threshold=1e-3, iterations=None):
    """The same pass from documents in HOST memory, the loop inside the library
    (``trlda_model_estep_corpus``, include/trlda_hip.h): ``offsets`` (int64, n_docs + 1), ``ids``,
    ``cnts`` (int32) are one CSR corpus -- what ``load_documents_csr`` / ``trlda_docs_from_text`` hand
    out; mini-batch i is documents [i * batch_size, (i + 1) * batch_size).  ``gamma0`` / ``gamma``:
    K x n_docs on the device; the statistics of mini-batch i go to ``sstats_ring[i % len(sstats_ring)]``
    (at least three K x V device arrays).  Batches are indexed and uploaded by the library's worker
    threads eight steps ahead of their E-step.  Returns when everything is enqueued; the results
    are complete on the model's stream (``model.lambdas`` or ``EStepStream.synchronize`` wait)."""
    import numpy as np
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    cnts = np.ascontiguousarray(cnts, dtype=np.int32)
    n_docs = len(offsets) - 1
    if n_docs < 0 or len(ids) != offsets[-1] or len(cnts) != offsets[-1]:
        raise TypeError("offsets / ids / cnts do not describe one CSR corpus.")
    if len(sstats_ring) < 3:
        raise ValueError("at least three statistics arrays (two E-steps are in flight).")
    K, V = model.num_topics, model.num_words
    dev = getattr(model, "_device", None)
    ring = (C.c_void_p * len(sstats_ring))(*[_address(s, "sstats", K * V, device=dev) for s in sstats_ring])
    _ffi.check(_ffi.lib().trlda_model_estep_corpus(
        model._handle, n_docs, offsets.ctypes.data, ids.ctypes.data, cnts.ctypes.data, int(batch_size),
        _address(gamma0, "gamma0", K * n_docs, device=dev), _address(gamma, "gamma", K * n_docs, device=dev),
        ring, len(sstats_ring), int(max_iter), float(threshold),
        _address(iterations, "iterations", n_docs, dtype="int32", device=dev)))
