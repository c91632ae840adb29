"""ctypes bindings for the CHECKERS under oracle/ -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing in trlda_amd/ does (tests/test_boundary.py greps for it).

* ``Oracle``    -> oracle/liboracle.so      (plain-C restatement, cpu_ref.c)
* ``Reference`` -> oracle/_ref/libtrlda_ref.so (the reference's own C++ core behind
  oracle/ref_shim.cpp); present only when built in a container that has
  /root/reference.  ``Reference.available()`` says whether it is.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="F_CONTIGUOUS")


def build(ref=True):
    """Run oracle/Makefile (gcc; and g++ over /root/reference when it exists)."""
    subprocess.run(["make", "-C", _HERE, "liboracle.so"] + (["ref"] if ref else []),
                   check=True, stdout=subprocess.DEVNULL)


def _csr(indptr, ids, cnts):
    return (np.ascontiguousarray(indptr, np.int32), np.ascontiguousarray(ids, np.int32),
            np.ascontiguousarray(cnts, np.int32))


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


class Oracle:
    """oracle/cpu_ref.c"""

    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.lib = C.CDLL(path)
        L.oracle_digamma.restype = C.c_double
        L.oracle_digamma.argtypes = [C.c_double]
        L.oracle_seed.argtypes = [C.c_uint]
        L.oracle_sample_gamma.argtypes = [C.c_int, C.c_int, C.c_int, _f64p]
        L.oracle_exp_elog_beta.argtypes = [C.c_int, C.c_int, _f64p, _f64p]
        est = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _f64p, _f64p, _f64p, _f64p,
               C.c_int, C.c_double, C.c_void_p]
        L.oracle_estep.argtypes = est
        L.oracle_estep_mt.argtypes = est + [C.c_int]
        L.oracle_tr_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                     _i32p, _i32p, _i32p, _f64p, _f64p]
        L.oracle_mstep_blend.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                         _f64p, _f64p, _f64p]
        L.oracle_online_update_parameters.restype = C.c_double
        L.oracle_online_update_parameters.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _f64p, _f64p, C.c_double,
            C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double,
            C.POINTER(C.c_int), C.c_void_p]
        L.oracle_batch_update_parameters.restype = C.c_double
        L.oracle_batch_update_parameters.argtypes = [
            C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _f64p, _f64p, C.c_double,
            C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]

    def digamma(self, x):
        return self.lib.oracle_digamma(float(x))

    def seed(self, s):
        self.lib.oracle_seed(int(s))

    def sample_gamma(self, m, n, k):
        out = np.zeros((m, n), order="F")
        self.lib.oracle_sample_gamma(m, n, k, out)
        return out

    def exp_elog_beta(self, lam):
        lam = _f(lam)
        out = np.zeros_like(lam, order="F")
        self.lib.oracle_exp_elog_beta(lam.shape[0], lam.shape[1], lam, out)
        return out

    def estep(self, lam, alpha, indptr, ids, cnts, gamma0, max_iter=100, threshold=1e-3,
              nthreads=0):
        """-> gamma (K,B), sstats (K,V), iters (B,)"""
        lam = _f(lam)
        K, V = lam.shape
        indptr, ids, cnts = _csr(indptr, ids, cnts)
        B = len(indptr) - 1
        gamma = _f(np.array(gamma0, dtype=np.float64, copy=True).reshape(K, B))
        sstats = np.zeros((K, V), order="F")
        iters = np.zeros(B, np.int32)
        alpha = _f(np.broadcast_to(np.asarray(alpha, np.float64).ravel(), (K,)).copy())
        args = [K, V, B, indptr, ids, cnts, lam, alpha, gamma, sstats, int(max_iter),
                float(threshold), iters.ctypes.data]
        if nthreads:
            rc = self.lib.oracle_estep_mt(*args, int(nthreads))
        else:
            rc = self.lib.oracle_estep(*args)
        if rc != 0:
            raise RuntimeError("oracle_estep: word id out of range")
        return gamma, sstats, iters

    def lower_bound(self, lam, alpha, eta, indptr, ids, cnts, gamma, sstats, factor=1.,
                    reference_indexing=False):
        """LDA::lowerBound (lda.cpp:304-360) from the E-step's gamma and sstats"""
        lam = _f(lam)
        K, V = lam.shape
        indptr, ids, cnts = _csr(indptr, ids, cnts)
        alpha = _f(np.broadcast_to(np.asarray(alpha, np.float64).ravel(), (K,)).copy())
        f = self.lib.oracle_lower_bound
        f.restype = C.c_double
        f.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _f64p, _f64p, C.c_double,
                      _f64p, _f64p, C.c_double, C.c_int]
        return f(K, V, len(indptr) - 1, indptr, ids, cnts, lam, alpha, float(eta), _f(gamma),
                 _f(sstats), float(factor), int(reference_indexing))

    def tr_init(self, lam_prime, indptr, ids, cnts, D, rho, eta):
        lam_prime = _f(lam_prime)
        K, V = lam_prime.shape
        indptr, ids, cnts = _csr(indptr, ids, cnts)
        out = np.zeros_like(lam_prime, order="F")
        self.lib.oracle_tr_init(K, V, len(indptr) - 1, int(D), rho, eta, indptr, ids, cnts,
                                lam_prime, out)
        return out

    def mstep_blend(self, lam_prime, sstats, rho, eta, scale):
        lam_prime = _f(lam_prime)
        out = np.zeros_like(lam_prime, order="F")
        self.lib.oracle_mstep_blend(lam_prime.shape[0], lam_prime.shape[1], rho, eta, scale,
                                    lam_prime, _f(sstats), out)
        return out

    def online_update_parameters(self, lam, alpha, eta, D, indptr, ids, cnts, update_count,
                                 max_iter_tr=10, max_iter_inference=20, kappa=.7, tau=100.,
                                 rho=-1., init_gamma=True, update_lambda=True, threshold=1e-3):
        """-> (rho, new lambda, new update_count, last gamma)"""
        lam = _f(np.array(lam, copy=True))
        K, V = lam.shape
        indptr, ids, cnts = _csr(indptr, ids, cnts)
        B = len(indptr) - 1
        alpha = _f(np.broadcast_to(np.asarray(alpha, np.float64).ravel(), (K,)).copy())
        cnt = C.c_int(int(update_count))
        gamma = np.zeros((K, max(B, 1)), order="F")
        r = self.lib.oracle_online_update_parameters(
            K, V, B, int(D), indptr, ids, cnts, lam, alpha, float(eta), int(max_iter_tr),
            int(max_iter_inference), float(kappa), float(tau), float(rho), int(init_gamma),
            int(update_lambda), float(threshold), C.byref(cnt), gamma.ctypes.data)
        return r, lam, cnt.value, gamma[:, :B]

    def batch_update_parameters(self, lam, alpha, eta, indptr, ids, cnts, max_epochs=100,
                                max_iter_inference=100, update_lambda=True, threshold=1e-3):
        lam = _f(np.array(lam, copy=True))
        K, V = lam.shape
        indptr, ids, cnts = _csr(indptr, ids, cnts)
        B = len(indptr) - 1
        alpha = _f(np.broadcast_to(np.asarray(alpha, np.float64).ravel(), (K,)).copy())
        gamma = np.zeros((K, max(B, 1)), order="F")
        r = self.lib.oracle_batch_update_parameters(
            K, V, B, indptr, ids, cnts, lam, alpha, float(eta), int(max_epochs),
            int(max_iter_inference), int(update_lambda), float(threshold), gamma.ctypes.data)
        return r, lam, gamma[:, :B]


class Reference:
    """oracle/_ref/libtrlda_ref.so -- the reference's own C++ (see ref_shim.cpp)."""

    @staticmethod
    def path(omp=False):
        return os.path.join(_HERE, "_ref", "libtrlda_ref_omp.so" if omp else "libtrlda_ref.so")

    @classmethod
    def available(cls, omp=False):
        return os.path.exists(cls.path(omp))

    def __init__(self, omp=False):
        L = self.lib = C.CDLL(self.path(omp))
        L.ref_last_error.restype = C.c_char_p
        L.ref_seed.argtypes = [C.c_uint]
        L.ref_digamma.restype = C.c_double
        L.ref_digamma.argtypes = [C.c_double]
        L.ref_polygamma.restype = C.c_double
        L.ref_polygamma.argtypes = [C.c_int, C.c_double]
        L.ref_sample_gamma.argtypes = [C.c_int, C.c_int, C.c_int, _f64p]
        L.ref_online_create.restype = C.c_void_p
        L.ref_online_create.argtypes = [C.c_int, C.c_int, C.c_int, _f64p, C.c_double]
        L.ref_batch_create.restype = C.c_void_p
        L.ref_batch_create.argtypes = [C.c_int, C.c_int, _f64p, C.c_double]
        L.ref_cumulative_create.restype = C.c_void_p
        L.ref_cumulative_create.argtypes = [C.c_int, C.c_int, _f64p, C.c_double]
        L.ref_batch_update_parameters_full.restype = C.c_double
        L.ref_batch_update_parameters_full.argtypes = [
            C.c_void_p, C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.ref_cumulative_update_parameters.restype = C.c_double
        L.ref_cumulative_update_parameters.argtypes = [
            C.c_void_p, C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_double, C.c_double, C.c_double]
        L.ref_model_destroy.argtypes = [C.c_void_p]
        L.ref_model_get_lambda.argtypes = [C.c_void_p, _f64p]
        L.ref_model_set_lambda.argtypes = [C.c_void_p, C.c_int, C.c_int, _f64p]
        L.ref_model_get_alpha.argtypes = [C.c_void_p, _f64p]
        L.ref_model_set_alpha.argtypes = [C.c_void_p, C.c_int, _f64p]
        L.ref_model_get_eta.restype = C.c_double
        L.ref_model_get_eta.argtypes = [C.c_void_p]
        L.ref_online_update_count.argtypes = [C.c_void_p]
        L.ref_model_estep.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _i32p, C.c_int, _f64p,
                                      _f64p, C.c_int, C.c_double]
        L.ref_model_lower_bound.restype = C.c_double
        L.ref_model_lower_bound.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, _i32p, _i32p,
                                            C.c_int, C.c_int]
        L.ref_online_update_parameters.restype = C.c_double
        L.ref_online_update_parameters.argtypes = [
            C.c_void_p, C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_double, C.c_double,
            C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.ref_batch_update_parameters.restype = C.c_double
        L.ref_batch_update_parameters.argtypes = [
            C.c_void_p, C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]

    def seed(self, s):
        self.lib.ref_seed(int(s))

    def digamma(self, x):
        return self.lib.ref_digamma(float(x))

    def polygamma(self, n, x):
        return self.lib.ref_polygamma(int(n), float(x))

    def sample_gamma(self, m, n, k):
        out = np.zeros((m, n), order="F")
        self.lib.ref_sample_gamma(m, n, k, out)
        return out

    def online(self, V, K, D, alpha=.1, eta=.3):
        return RefModel(self, "online", V, K, D, alpha, eta)

    def batch(self, V, K, alpha=.1, eta=.3):
        return RefModel(self, "batch", V, K, 0, alpha, eta)

    def cumulative(self, V, K, alpha=.1, eta=.3):
        return RefModel(self, "cumulative", V, K, 0, alpha, eta)


class RefModel:
    def __init__(self, ref, kind, V, K, D, alpha, eta):
        self.ref, self.kind, self.V, self.K = ref, kind, V, K
        a = _f(np.broadcast_to(np.asarray(alpha, np.float64).ravel(), (K,)).copy())
        if kind == "online":
            self.h = ref.lib.ref_online_create(V, K, D, a, float(eta))
        elif kind == "cumulative":
            self.h = ref.lib.ref_cumulative_create(V, K, a, float(eta))
        else:
            self.h = ref.lib.ref_batch_create(V, K, a, float(eta))
        if not self.h:
            raise RuntimeError(ref.lib.ref_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.ref.lib.ref_model_destroy(self.h)
            self.h = None

    @property
    def lambdas(self):
        out = np.zeros((self.K, self.V), order="F")
        self.ref.lib.ref_model_get_lambda(self.h, out)
        return out

    @lambdas.setter
    def lambdas(self, lam):
        lam = _f(lam)
        if self.ref.lib.ref_model_set_lambda(self.h, lam.shape[0], lam.shape[1], lam) != 0:
            raise RuntimeError(self.ref.lib.ref_last_error().decode())

    @property
    def alpha(self):
        out = np.zeros((self.K,), order="F")
        self.ref.lib.ref_model_get_alpha(self.h, out)
        return out

    @alpha.setter
    def alpha(self, a):
        a = _f(np.asarray(a, np.float64).ravel())
        if self.ref.lib.ref_model_set_alpha(self.h, len(a), a) != 0:
            raise RuntimeError(self.ref.lib.ref_last_error().decode())

    @property
    def eta(self):
        return self.ref.lib.ref_model_get_eta(self.h)

    @property
    def update_count(self):
        return self.ref.lib.ref_online_update_count(self.h)

    def estep(self, indptr, ids, cnts, gamma0=None, max_iter=100, threshold=1e-3):
        indptr, ids, cnts = _csr(indptr, ids, cnts)
        B = len(indptr) - 1
        if gamma0 is None:
            gamma = np.zeros((self.K, B), order="F")
        else:
            gamma = _f(np.array(gamma0, dtype=np.float64, copy=True).reshape(self.K, B))
        sstats = np.zeros((self.K, self.V), order="F")
        rc = self.ref.lib.ref_model_estep(self.h, B, indptr, ids, cnts, int(gamma0 is not None),
                                          gamma, sstats, int(max_iter), float(threshold))
        if rc != 0:
            raise RuntimeError(self.ref.lib.ref_last_error().decode())
        return gamma, sstats

    def lower_bound(self, indptr, ids, cnts, num_documents=-1, max_iter=100):
        """LDA::lowerBound (lda.cpp:297-360); gamma0 from the seeded libc stream"""
        indptr, ids, cnts = _csr(indptr, ids, cnts)
        return self.ref.lib.ref_model_lower_bound(self.h, int(self.kind == "online"),
                                                  len(indptr) - 1, indptr, ids, cnts,
                                                  int(num_documents), int(max_iter))

    def update_parameters(self, indptr, ids, cnts, max_iter_tr=10, max_iter_inference=20,
                          kappa=.7, tau=100., rho=-1., adaptive=False, init_gamma=True,
                          update_lambda=True, update_alpha=False, update_eta=False,
                          min_alpha=1e-6, min_eta=1e-6, max_epochs=100, max_iter_alpha=10,
                          max_iter_eta=20, emp_bayes_threshold=1e-8, inference_threshold=1e-3):
        indptr, ids, cnts = _csr(indptr, ids, cnts)
        B = len(indptr) - 1
        if self.kind == "cumulative":
            return self.ref.lib.ref_cumulative_update_parameters(
                self.h, B, indptr, ids, cnts, max_epochs, max_iter_inference, max_iter_alpha,
                int(update_lambda), int(update_alpha), min_alpha, emp_bayes_threshold,
                inference_threshold)
        if self.kind == "batch":
            return self.ref.lib.ref_batch_update_parameters_full(
                self.h, B, indptr, ids, cnts, max_epochs, max_iter_inference, max_iter_alpha,
                max_iter_eta, int(update_lambda), int(update_alpha), int(update_eta), min_alpha,
                min_eta, emp_bayes_threshold)
        if self.kind == "online":
            return self.ref.lib.ref_online_update_parameters(
                self.h, B, indptr, ids, cnts, max_iter_tr, max_iter_inference, kappa, tau, rho,
                int(adaptive), int(init_gamma), int(update_lambda), int(update_alpha),
                int(update_eta), min_alpha, min_eta)
        return self.ref.lib.ref_batch_update_parameters(
            self.h, B, indptr, ids, cnts, max_epochs, max_iter_inference, int(update_lambda),
            int(update_alpha), int(update_eta))
