"""``trlda_amd.models`` -- host-side mirror of ``trlda.models`` (reference
python/models/__init__.py:1-5) for the accelerated path.

Same class names, constructor arguments, properties, method signatures, defaults and
error behaviour as the reference's CPython types (python/src/module.cpp,
ldainterface.cpp, onlineldainterface.cpp, batchldainterface.cpp); the computation
goes through ``libtrlda_hip.so`` (include/trlda_hip.h) to the gfx950 kernels.  State
(lambda) lives in HBM; getters return fresh, read-only, Fortran-ordered copies just
like ``PyArray_FromMatrixXd`` (python/src/pyutils.cpp:15-36).
"""
import ctypes as C
import os

import numpy as np

from .. import _ffi
from ..documents import CSRDocuments, DeviceBatch, as_csr

__all__ = ["Distribution", "LDA", "OnlineLDA", "BatchLDA", "CumulativeLDA"]


def _default_device():
    return int(os.environ.get("LOCAL_RANK", "0"))


def _alpha_vector(alpha, num_topics):
    """alpha argument handling of OnlineLDA_init (onlineldainterface.cpp:58-83).

    Returns (K, alpha[K]).  A scalar keeps ``num_topics``; an array *defines* K
    (the reference calls the ArrayXd constructor and ignores num_topics)."""
    if alpha is None:
        alpha = .1
    if isinstance(alpha, (float, int, np.floating, np.integer)) and not isinstance(alpha, bool):
        return int(num_topics), np.full(int(num_topics), float(alpha), dtype=np.float64)
    try:
        arr = np.asarray(alpha, dtype=np.float64)
    except (TypeError, ValueError):
        raise TypeError("Alpha should be of type `ndarray`.")
    if arr.ndim == 0:
        return int(num_topics), np.full(int(num_topics), float(arr), dtype=np.float64)
    if arr.ndim == 1:
        arr = arr.reshape(-1, 1)
    if arr.ndim != 2:
        raise TypeError("Alpha should be one-dimensional.")
    if arr.shape[0] == 1:
        arr = arr.T
    if arr.shape[1] != 1:
        raise TypeError("Alpha should be one-dimensional.")
    return arr.shape[0], np.ascontiguousarray(arr[:, 0])


def _inference_method(name):
    """ldainterface.cpp:343-359: first letter decides; returns 'VI' or 'GIBBS'."""
    if name is None:
        return "VI"
    if not isinstance(name, str):
        raise TypeError("`inference_method` should be either 'GIBBS' or 'VI'.")
    first = name[:1]
    if first in ("v", "V"):
        return "VI"
    if first in ("g", "G"):
        return "GIBBS"
    raise TypeError("`inference_method` should be either 'GIBBS' or 'VI'.")


class Distribution(object):
    """Abstract base (reference include/distribution.h, distributioninterface.cpp)."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("This is an abstract class.")


class LDA(Distribution):
    """Base of the LDA models: state + the E-step (reference include/lda.h)."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("This is an abstract class.")      # ldainterface.cpp:35-38

    # -- construction shared by the subclasses --------------------------------------
    def _setup(self, num_words, num_topics, alpha, eta, device, _lambda=None):
        if int(num_words) <= 0:
            raise RuntimeError("Number of words should be positive.")
        K, alpha_vec = _alpha_vector(alpha, num_topics)
        if K <= 0:
            raise RuntimeError("Number of topics should be positive.")
        self._V = int(num_words)
        self._K = int(K)
        self._alpha = alpha_vec
        self._eta = float(eta)
        self._device = _default_device() if device is None else int(device)
        self._handle = _ffi.vp()
        L = _ffi.lib()
        _ffi.require_gpu()
        _ffi.check(L.trlda_model_create(C.byref(self._handle), self._device, self._K, self._V))
        _ffi.check(L.trlda_model_set_alpha(self._handle, self._alpha))
        if _lambda is None:
            # lambda = sampleGamma(K, V, 100) / 100 from libc rand()       (lda.cpp:71)
            lam = np.empty((self._K, self._V), dtype=np.float64, order="F")
            L.trlda_sample_gamma_init(self._K, self._V, lam)
        else:
            lam = np.asfortranarray(_lambda, dtype=np.float64)
        _ffi.check(L.trlda_model_set_lambda(self._handle, lam))

    # An empirical-Bayes step whose device sums are still on their way (OnlineLDA defers the wait
    # so that the next mini-batch is parsed, converted and uploaded meanwhile): (rho, min_alpha,
    # min_eta), finished by whoever next needs alpha, eta or an E-step.
    _eb_pending = None

    def _settle(self):
        pending = self._eb_pending
        if pending is None:
            return
        self._eb_pending = None
        rho, min_alpha, min_eta = pending
        alpha = np.ascontiguousarray(self._alpha, dtype=np.float64).copy()
        eta = C.c_double(self._eta)
        _ffi.check(_ffi.lib().trlda_model_online_eb_finish(self._handle, rho, min_alpha, min_eta,
                                                          alpha, C.byref(eta)))
        self._alpha, self._eta = alpha, float(eta.value)

    def close(self):
        # an empirical-Bayes step still on its way is finished, not dropped: alpha / eta stay
        # readable after close() with the values the last update_parameters call gave them
        if getattr(self, "_handle", None) and self._eb_pending is not None:
            try:
                self._settle()
            except Exception:                        # noqa: BLE001 -- closing must not raise
                self._eb_pending = None
        self._eb_pending = None
        if getattr(self, "_handle", None):
            _ffi.lib().trlda_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- properties (ldainterface.cpp:41-148, module.cpp:67-90) ---------------------
    @property
    def num_topics(self):
        return self._K

    @property
    def num_words(self):
        return self._V

    @property
    def device(self):
        return self._device

    def _get_lambda(self):
        lam = np.empty((self._K, self._V), dtype=np.float64, order="F")
        _ffi.check(_ffi.lib().trlda_model_get_lambda(self._handle, lam))
        lam.flags.writeable = False                                  # ldainterface.cpp:57
        return lam

    def _set_lambda(self, value):
        try:
            arr = np.asarray(value, dtype=np.float64)
        except (TypeError, ValueError):
            raise TypeError("Lambda should be of type `ndarray`.")
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)                                 # pyutils.cpp:91-128
        if arr.ndim != 2 or arr.shape != (self._K, self._V):
            raise RuntimeError("Lambda has wrong dimensionality.")   # lda.h:186-187
        _ffi.check(_ffi.lib().trlda_model_set_lambda(self._handle, np.asfortranarray(arr)))

    lambdas = property(_get_lambda, _set_lambda,
                       doc="Parameters governing beliefs over topics (K x V).")
    _lambda = property(_get_lambda, _set_lambda, doc="Alias for `lambdas`.")

    @property
    def alpha(self):
        self._settle()
        return self._alpha.reshape(-1, 1).copy(order="F")            # K x 1, ldainterface.cpp:87

    @alpha.setter
    def alpha(self, value):
        if isinstance(value, (float, int, np.floating, np.integer)) and \
                not isinstance(value, bool):
            if value < 0.:
                raise RuntimeError("Alpha should not be negative.")  # lda.h:147-151
            new = np.full(self._K, float(value), dtype=np.float64)
        else:
            _, new = _alpha_vector(value, self._K)
            if new.size != self._K:
                raise RuntimeError("Alpha has wrong dimensionality.")  # lda.h:155-156
            if (new < 0.).any():
                raise RuntimeError("Alpha should not be negative.")
        self._settle()
        _ffi.check(_ffi.lib().trlda_model_set_alpha(self._handle, new))
        self._alpha = new

    @property
    def eta(self):
        self._settle()
        return self._eta

    @eta.setter
    def eta(self, value):
        value = float(value)
        if value < 0.:
            raise RuntimeError("Eta should not be negative.")        # lda.h:172-173
        self._settle()
        self._eta = value

    # -- documents ---------------------------------------------------------------
    def upload(self, docs):
        """Convert + upload a batch once; the result can be passed as ``docs``."""
        if isinstance(docs, DeviceBatch):
            return docs
        return DeviceBatch(docs, self._V, self._device)

    def _batch(self, docs):
        if isinstance(docs, DeviceBatch):
            if docs.num_words != self._V or docs.device != self._device:
                raise RuntimeError("Batch was uploaded for a different model.")
            return docs, False
        return DeviceBatch(docs, self._V, self._device), True

    # -- E-step (ldainterface.cpp:311-390 -> lda.cpp:119-220) ------------------------
    def update_variables(self, docs, latents=None, inference_method='VI', max_iter=100,
                         threshold=0.001, num_samples=1, burn_in=2, return_iterations=False):
        """E-step: returns ``(gamma K x N, sstats K x V)`` as Fortran-ordered float64."""
        method = _inference_method(inference_method)
        if method != "VI":
            raise NotImplementedError(
                "Gibbs inference (lda.cpp:224-293) is outside the accelerated path.")
        batch, owned = self._batch(docs)
        try:
            self._settle()
            B = len(batch)
            L = _ffi.lib()
            if latents is not None:
                try:
                    g = np.array(latents, dtype=np.float64, order="F", copy=True)
                except (TypeError, ValueError):
                    raise TypeError("`latents` should be of type `ndarray`.")
                if g.ndim == 1:
                    g = g.reshape(-1, 1, order="F")
                if g.ndim != 2 or g.shape != (self._K, B):
                    raise RuntimeError("Initial gamma has wrong dimensionality.")  # lda.cpp:165
                gamma = np.asfortranarray(g)
            else:
                gamma = np.empty((self._K, B), dtype=np.float64, order="F")
                L.trlda_sample_gamma_init(self._K, B, gamma)          # lda.cpp:135
            sstats = np.empty((self._K, self._V), dtype=np.float64, order="F")
            iters = np.zeros(B, dtype=np.int32)
            _ffi.check(L.trlda_model_estep_host(self._handle, batch.handle, gamma, sstats,
                                                int(max_iter), float(threshold),
                                                iters.ctypes.data))
        finally:
            if owned:
                batch.close()
        if return_iterations:
            return gamma, sstats, iters
        return gamma, sstats

    do_e_step = update_variables                                     # module.cpp:103-106

    # -- the variational lower bound (ldainterface.cpp:394-470 -> lda.cpp:297-360) ----------
    def _default_num_documents(self):
        return -1                                                    # lda.h: numDocuments = -1

    def lower_bound(self, docs, num_documents=-1, inference_method='VI', max_iter=100,
                    num_samples=1, burn_in=2):
        """Estimate of the lower bound on the given documents (fresh E-step from a random
        gamma drawn from the seeded libc stream, as the reference does), scaled to
        ``num_documents`` when that is given.

        Deviation from upstream's actual output: the reference's src/lda.cpp:334 reads
        ``psiLambda.row(id)`` of the K x V matrix where the word's column is meant (an indexing
        slip its release build does not trap); this implements the column read -- the formula of
        the paper and of Hoffman's ``approx_bound`` -- so values differ from upstream's by about
        1e-4 relative on its own test set-up (the oracle can reproduce either)."""
        method = _inference_method(inference_method)
        if method != "VI":
            raise NotImplementedError(
                "Gibbs inference (lda.cpp:224-293) is outside the accelerated path.")
        num_documents = int(num_documents)
        if num_documents < 0:
            num_documents = self._default_num_documents()            # onlinelda.cpp:184-191
        batch, owned = self._batch(docs)
        try:
            self._settle()
            B = len(batch)
            if B == 0:
                raise RuntimeError("The lower bound needs at least one document.")
            L = _ffi.lib()
            gamma = np.empty((self._K, B), dtype=np.float64, order="F")
            L.trlda_sample_gamma_init(self._K, B, gamma)              # lda.cpp:309 -> :135
            factor = num_documents / float(B) if num_documents >= 0 else 1.   # lda.cpp:302-303
            bound = C.c_double(0.)
            _ffi.check(L.trlda_model_lower_bound(self._handle, batch.handle, gamma, self._eta,
                                                 factor, int(max_iter), 0.001, C.byref(bound)))
        finally:
            if owned:
                batch.close()
        return bound.value

    # -- device reductions for the empirical-Bayes steps (csrc/eb_kernels.h) ----------------
    def _psi_gamma_diff_device(self, num_docs):
        """sum_d (psi(gamma_dk) - psi(sum_k gamma_dk)) over the gamma the last update / resident
        E-step left on the device (onlinelda.cpp:123-128, batchlda.cpp:72-74): K numbers."""
        out = np.empty(self._K, dtype=np.float64)
        _ffi.check(_ffi.lib().trlda_model_eb_gamma_stats(self._handle, int(num_docs), None, out))
        return out

    def _lambda_psi_stats_device(self):
        """(sum_kw psi(lambda_kw), row sums of lambda): onlinelda.cpp:152-154, batchlda.cpp:152."""
        total = C.c_double(0.)
        rowsums = np.empty(self._K, dtype=np.float64)
        _ffi.check(_ffi.lib().trlda_model_eb_lambda_stats(self._handle, C.byref(total), rowsums))
        return total.value, rowsums

    def _resident_estep(self, batch, max_iter, threshold=0.001):
        """updateVariables(documents, parameters) from a fresh random gamma, results left on the
        device (onlinelda.cpp:118-120)."""
        _ffi.check(_ffi.lib().trlda_model_estep_resident(self._handle, batch.handle, int(max_iter),
                                                         float(threshold)))

    # -- not on the accelerated path ----------------------------------------------
    def sample(self, num_documents, length):
        raise NotImplementedError("sample (lda.cpp:88-115) is outside the accelerated path.")

    def __str__(self):                                               # ldainterface.cpp:473-490
        self._settle()
        return "Number of topics: %d\nEta: %.4g\nAlpha: %.4g, %.4g (min, max)\n" % (
            self._K, self._eta, self._alpha.min(), self._alpha.max())


class OnlineLDA(LDA):
    """Online trust-region LDA (reference src/onlinelda.cpp, onlineldainterface.cpp).

        >>> model = OnlineLDA(num_words=7000, num_topics=100, num_documents=10000,
        ...                   alpha=.1, eta=.3)

    ``alpha`` can be a scalar or an array with one entry for each topic.
    """

    def __init__(self, num_words, num_topics, num_documents, alpha=.1, eta=.3, kappa_=0.,
                 tau_=0., device=None):
        # kappa_ / tau_ are accepted and ignored (old pickles; onlineldainterface.cpp:50-52)
        self._num_documents = int(num_documents)
        self._update_count = 0
        # adaptive learning rate state (onlinelda.cpp:28-31; not pickled, like the reference)
        self._ada_tau = 1000.
        self._ada_rho = 1. / self._ada_tau
        self._ada_sq_norm = 1.
        self._setup(num_words, num_topics, alpha, eta, device)

    def _default_num_documents(self):
        return self._num_documents                                   # onlinelda.cpp:184-191

    @property
    def num_documents(self):
        return self._num_documents

    @num_documents.setter
    def num_documents(self, value):
        value = int(value)
        if value < 0:                                                # onlinelda.h:57-58
            raise RuntimeError("The number of documents should not be negative.")
        self._num_documents = value

    @property
    def update_count(self):
        return self._update_count

    @update_count.setter
    def update_count(self, value):
        value = int(value)
        if value < 0:                                                # onlinelda.h:71-72
            raise RuntimeError("The update count should not be negative.")
        self._update_count = value

    def update_parameters(self, docs, max_iter_tr=10, max_iter_inference=20, kappa=.7,
                          tau=100., rho=-1., adaptive=False, init_gamma=True,
                          update_lambda=True, update_alpha=False, update_eta=False,
                          min_alpha=1e-6, min_eta=1e-6, verbosity=0):
        """One online update; returns the learning rate used
        (onlineldainterface.cpp:204-256 -> onlinelda.cpp:53-179).

        The lambda path (E-steps, trust-region loop, M-steps) runs on the GPU, and so do the
        sums over gamma, lambda and the statistics that the empirical-Bayes steps for alpha and
        eta and the adaptive learning rate need (onlinelda.cpp:116-175, csrc/eb_kernels.h); the
        host keeps the K- and scalar-sized Newton steps."""
        batch, owned = self._batch(docs)
        try:
            # (the previous call's empirical-Bayes step, if it is still on its way: the conversion
            # and upload above ran beside the device's work on that call)
            self._settle()
            B = len(batch)
            if B == 0:
                return 1.0                                           # onlinelda.cpp:54-56
            L = _ffi.lib()
            rho_arg = float(rho)
            if rho_arg < 0. and adaptive:
                rho_arg = self._ada_rho                              # onlinelda.cpp:61-62
            eta_old = self._eta
            # the adaptive rate reads lambda' and the statistics after the update
            # (onlinelda.cpp:167-175): the device keeps both
            _ffi.check(L.trlda_model_set_keep_sstats(self._handle,
                                                     int(bool(adaptive and update_lambda))))
            count = C.c_int(self._update_count)
            rho_out = C.c_double(0.)
            _ffi.check(L.trlda_model_online_update(
                self._handle, batch.handle, self._num_documents, self._eta, int(max_iter_tr),
                int(max_iter_inference), float(kappa), float(tau), rho_arg,
                int(bool(init_gamma)), int(bool(update_lambda)), 0.001,  # lda.h:56: fixed
                C.byref(count), C.byref(rho_out), None))
            rho_used = rho_out.value

            if update_alpha or update_eta:                           # onlinelda.cpp:116-162
                if update_alpha and not update_lambda:
                    self._resident_estep(batch, max_iter_inference)
                # the device sums over gamma and lambda and their way back to the host are
                # enqueued; the wait, the K-sized Newton steps on the host and the new alpha's way
                # to the device are left to whoever next needs alpha, eta or an E-step (_settle):
                # a loop over mini-batches prepares its next one meanwhile
                _ffi.check(L.trlda_model_online_eb_begin(
                    self._handle, None, B, B, int(bool(update_alpha)), int(bool(update_eta))))
                self._eb_pending = (rho_used, float(min_alpha), float(min_eta))
                if update_lambda and adaptive:
                    self._settle()

            if update_lambda and adaptive:                           # onlinelda.cpp:167-175
                t = self._ada_tau
                u2, g2 = C.c_double(0.), C.c_double(0.)
                _ffi.check(L.trlda_model_adaptive_stats(
                    self._handle, eta_old, float(self._num_documents) / B, t,
                    C.byref(u2), C.byref(g2)))
                self._ada_sq_norm = (1. - 1. / t) * self._ada_sq_norm + 1. / t * u2.value
                self._ada_rho = g2.value / self._ada_sq_norm
                self._ada_tau = t * (1. - self._ada_rho) + 1.

            self._update_count = count.value
        finally:
            if owned:
                batch.close()
        return rho_used

    def __reduce__(self):                                            # onlineldainterface.cpp:265
        self._settle()
        args = (self._V, self._K, self._num_documents, self.alpha, self._eta)
        state = (self.lambdas, self._update_count)
        return (self.__class__, args, state)

    def __setstate__(self, state):                                   # onlineldainterface.cpp:296
        lam, count = state
        self.lambdas = lam
        self.update_count = count


def _online_alpha_step(alpha, psi_gamma_diff, num_docs, rho, min_alpha):
    """One natural-gradient step on alpha, onlinelda.cpp:123-142; psi_gamma_diff[k] = sum over the
    mini-batch's documents of psi(gamma_dk) - psi(sum_k gamma_dk).  K-sized host arithmetic in
    the library (csrc/eb_steps.cpp)."""
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    out = np.empty_like(alpha)
    _ffi.check(_ffi.lib().trlda_eb_online_alpha_step(
        alpha.size, alpha, np.ascontiguousarray(psi_gamma_diff, dtype=np.float64), float(num_docs),
        float(rho), float(min_alpha), out))
    return out


def _online_eta_step(eta, sum_psi_lambda, rowsums, K, V, rho, min_eta):
    """One Newton step on eta, onlinelda.cpp:147-162 (csrc/eb_steps.cpp)."""
    return float(_ffi.lib().trlda_eb_online_eta_step(
        float(eta), float(sum_psi_lambda), np.ascontiguousarray(rowsums, dtype=np.float64), int(K),
        int(V), float(rho), float(min_eta)))


class _Verbosity(object):
    """`verbosity > 1`: the line searches print their progress to stdout in the reference's words
    (batchlda.cpp:78-88,120-123,155-165,184-187; cumulativelda.cpp:87-97,129-132) -- from C, so
    Python's own buffer is flushed first to keep the order of what a caller prints around it."""

    def __init__(self, verbosity):
        self.verbosity = int(verbosity)

    def __enter__(self):
        if self.verbosity > 1:
            import sys
            sys.stdout.flush()
        _ffi.lib().trlda_eb_set_verbosity(self.verbosity)

    def __exit__(self, *exc):
        _ffi.lib().trlda_eb_set_verbosity(0)


def _eta_line_search(eta, sum_psi_lambda, rowsums, K, V, max_iter_eta, min_eta, threshold, verbosity=0):
    """Newton steps on eta with a step-halving line search on the lower bound,
    batchlda.cpp:147-205 (csrc/eb_steps.cpp)."""
    with _Verbosity(verbosity):
        return float(_ffi.lib().trlda_eb_eta_line_search(
            float(eta), float(sum_psi_lambda), np.ascontiguousarray(rowsums, dtype=np.float64), int(K),
            int(V), int(max_iter_eta), float(min_eta), float(threshold)))


def _alpha_line_search(alpha, psi_gamma_diff, num_docs, max_iter_alpha, min_alpha, threshold, verbosity=0):
    """Newton / natural-gradient steps on alpha with a step-halving line search on the lower
    bound: batchlda.cpp:81-141 == cumulativelda.cpp:90-150 (csrc/eb_steps.cpp)."""
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    out = np.empty_like(alpha)
    with _Verbosity(verbosity):
        _ffi.check(_ffi.lib().trlda_eb_alpha_line_search(
            alpha.size, alpha, np.ascontiguousarray(psi_gamma_diff, dtype=np.float64), float(num_docs),
            int(max_iter_alpha), float(min_alpha), float(threshold), out))
    return out


class BatchLDA(LDA):
    """Batch variational LDA (reference src/batchlda.cpp, batchldainterface.cpp)."""

    def __init__(self, num_words, num_topics, alpha=.1, eta=.3, device=None):
        self._setup(num_words, num_topics, alpha, eta, device)

    def update_parameters(self, docs, max_epochs=100, max_iter_inference=100, max_iter_alpha=10,
                          max_iter_eta=20, update_lambda=True, update_alpha=False,
                          update_eta=False, min_alpha=1e-6, min_eta=1e-6,
                          emp_bayes_threshold=1e-8, verbosity=0):
        """batchldainterface.cpp:126-172 -> batchlda.cpp:43-208.  The E-steps and
        lambda = eta + sstats run on the GPU, as do the sums over gamma and lambda behind the
        alpha / eta line searches (batchlda.cpp:66-205); the searches themselves are K- and
        scalar-sized and run on the host."""
        batch, owned = self._batch(docs)
        try:
            B = len(batch)
            if B == 0:
                return 1.                                            # batchlda.cpp:44-46
            L = _ffi.lib()
            K, V = self._K, self._V
            if not (update_alpha or update_eta):
                _ffi.check(L.trlda_model_batch_update(
                    self._handle, batch.handle, self._eta, int(max_epochs),
                    int(max_iter_inference), int(bool(update_lambda)), 0.001, None))
                return 1.
            for _epoch in range(int(max_epochs)):                    # batchlda.cpp:48
                if update_lambda:
                    _ffi.check(L.trlda_model_batch_update(
                        self._handle, batch.handle, self._eta, 1, int(max_iter_inference), 1,
                        0.001, None))
                if update_alpha:                                     # batchlda.cpp:64-142
                    if not update_lambda:
                        self._resident_estep(batch, max_iter_inference)
                    alpha = _alpha_line_search(self._alpha, self._psi_gamma_diff_device(B), B,
                                               max_iter_alpha, min_alpha, emp_bayes_threshold, verbosity)
                    _ffi.check(L.trlda_model_set_alpha(self._handle, np.ascontiguousarray(alpha)))
                    self._alpha = alpha
                if update_eta:                                       # batchlda.cpp:147-205
                    sum_psi, rowsums = self._lambda_psi_stats_device()
                    self._eta = _eta_line_search(self._eta, sum_psi, rowsums, K, V, max_iter_eta,
                                                 min_eta, emp_bayes_threshold, verbosity)
        finally:
            if owned:
                batch.close()
        return 1.                                                    # batchlda.cpp:207

    def __reduce__(self):                                            # batchldainterface.cpp:181
        return (self.__class__, (self._V, self._K, self.alpha, self._eta), (self.lambdas,))

    def __setstate__(self, state):
        self.lambdas = state[0]


class CumulativeLDA(LDA):
    """SDA-Bayes streaming LDA (reference src/cumulativelda.cpp, cumulativeldainterface.cpp).

        >>> model = CumulativeLDA(num_words=7000, num_topics=100, alpha=.1, eta=.3)
        >>> for documents in load_documents('data_train.dat', 1000):
        ...     model.update_parameters(documents, max_epochs=100)

    In contrast to OnlineLDA, each document should be processed only once.
    """

    def __init__(self, num_words, num_topics, alpha=.1, eta=.3, device=None):
        # the base constructor draws a random lambda from the libc stream (lda.cpp:71) before
        # CumulativeLDA overwrites it with eta (cumulativelda.cpp:30): keep the stream in step
        self._setup(num_words, num_topics, alpha, eta, device)
        self.lambdas = np.full((self._K, self._V), float(eta))
        self._psi_gamma_diff = np.zeros(self._K)
        self._num_documents = 0

    def update_parameters(self, docs, max_epochs=100, max_iter_inference=100, max_iter_alpha=10,
                          update_lambda=True, update_alpha=False, min_alpha=1e-6,
                          emp_bayes_threshold=1e-8, inference_threshold=0.001, verbosity=0):
        """cumulativeldainterface.cpp:115-160 -> cumulativelda.cpp:49-153."""
        batch, owned = self._batch(docs)
        try:
            B = len(batch)
            if B == 0:
                return 1.                                            # cumulativelda.cpp:50-52
            L = _ffi.lib()
            _ffi.check(L.trlda_model_cumulative_update(
                self._handle, batch.handle, int(max_epochs), int(max_iter_inference),
                int(bool(update_lambda)), float(inference_threshold), None))
            if update_alpha:                                         # cumulativelda.cpp:76-150
                self._resident_estep(batch, max_iter_inference, inference_threshold)
                self._psi_gamma_diff = self._psi_gamma_diff + self._psi_gamma_diff_device(B)
                self._num_documents += B
                alpha = _alpha_line_search(self._alpha, self._psi_gamma_diff, self._num_documents,
                                           max_iter_alpha, min_alpha, emp_bayes_threshold, verbosity)
                _ffi.check(L.trlda_model_set_alpha(self._handle, np.ascontiguousarray(alpha)))
                self._alpha = alpha
        finally:
            if owned:
                batch.close()
        return 1.

    def __reduce__(self):                                            # cumulativeldainterface.cpp
        return (self.__class__, (self._V, self._K, self.alpha, self._eta), (self.lambdas,))

    def __setstate__(self, state):
        self.lambdas = state[0]
