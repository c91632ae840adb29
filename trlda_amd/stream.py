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
    # gamma[i] (B x K, a document's K values contiguous) and sstats[i] (V x K) are complete here

Consecutive calls must write different arrays (two are in flight); a caller that reuses one set
of arrays is recognised by the library and goes one call at a time.
"""
import ctypes as C

from . import _ffi

__all__ = ["EStepStream"]


def _address(x):
    """device address of a torch tensor (or anything with data_ptr()), or an int / None"""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        if hasattr(x, "is_contiguous") and not x.is_contiguous():
            raise TypeError("device arrays must be contiguous.")
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
        _ffi.check(self._lib.trlda_model_estep_io_ahead(
            self._model._handle, batch.handle, self._up, n, _address(gamma0), _address(gamma),
            _address(sstats), int(max_iter), float(threshold), _address(iterations)))
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

    def __exit__(self, *exc):
        self.close()
        return False
