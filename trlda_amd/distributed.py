"""Data-parallel OnlineLDA over the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm).

Two ways of meeting at the reference's reduction point (lda.cpp:211-217), chosen per call by the
bytes they move (``exchange="auto"``): the all-reduce of the K x V statistics described below,
and the FACTOR exchange of csrc/dp_kernels.h -- an all-gather of every document's expElogtheta
row and per-entry weights (8 (K + n_d) bytes per document) after which every rank forms the
whole mini-batch's statistics and the fused M-step itself, through
``trlda_model_online_update_dp`` on a RCCL communicator of the process's own (``rccl.py``).

Documents are independent given lambda (reference src/lda.cpp:176-214 touches only
column i of gamma and *adds* into sstats), so a mini-batch is split into contiguous
document ranges, one per rank.  lambda is replicated; every rank computes its own
preamble and its shard's sufficient statistics, the K x V statistics are summed with
ONE all-reduce where the reference has its ``omp critical`` reduction
(lda.cpp:211-217), and every rank then applies the identical M-step
(onlinelda.cpp:99-100), so lambda stays replicated without a broadcast.  The
trust-region initial step (onlinelda.cpp:79-86) needs the batch's word counts: one
more all-reduce of V integer-valued doubles per call.

The per-rank computation is behind a small *engine* interface so that the
sharding / collective / learning-rate logic can be exercised on CPU with the gloo
backend (tests/test_distributed_cpu.py plugs in a checker-backed engine there); the
product engine is :class:`HipEngine` and nothing else ships.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .documents import CSRDocuments, DeviceBatch, as_csr


class HipEngine(object):
    """Per-rank compute on one MI355X through libtrlda_hip.so; buffers are torch tensors
    (device memory + the objects RCCL reduces), kernels run on torch's current stream."""

    def __init__(self, num_words, num_topics, device):
        import torch
        self.torch = torch
        self.K, self.V = int(num_topics), int(num_words)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device)
        self.lib = _ffi.lib()
        self.handle = _ffi.vp()
        _ffi.check(self.lib.trlda_model_create(C.byref(self.handle), self.device_index,
                                               self.K, self.V))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _ffi.check(self.lib.trlda_model_set_stream(self.handle, _ffi.vp(stream)))
        kv = self.K * self.V
        self.lambda_prime = torch.empty(kv, dtype=torch.float64, device=self.device)
        self.sstats = torch.empty(kv, dtype=torch.float64, device=self.device)
        self.wc = torch.empty(self.V, dtype=torch.float64, device=self.device)
        self.gamma = None
        self.comm = None

    def make_communicator(self, dist, group):
        """A ncclComm_t of our own over the group's ranks (collective call); None if RCCL cannot
        be reached that way."""
        from . import rccl
        try:
            self.comm = rccl.own_communicator(dist, self.device, group)
        except Exception:                             # noqa: BLE001 -- the all-reduce path remains
            self.comm = None
        return self.comm

    def update_dp(self, batch, shard, cuts, rank, world, num_documents, eta, max_iter_tr,
                  max_iter_inference, kappa, tau, rho, init_gamma, threshold, update_count):
        """onlinelda.cpp:53-111 with the factor exchange; returns (rho, update_count)."""
        cuts = np.ascontiguousarray(cuts, dtype=np.int32)
        count, rho_out = C.c_int(int(update_count)), C.c_double(0.)
        _ffi.check(self.lib.trlda_model_online_update_dp(
            self.handle, batch.handle, shard.handle, self.comm, int(rank), int(world),
            cuts.ctypes.data_as(C.POINTER(C.c_int32)), int(num_documents), float(eta),
            int(max_iter_tr), int(max_iter_inference), float(kappa), float(tau), float(rho),
            int(bool(init_gamma)), float(threshold), C.byref(count), C.byref(rho_out)))
        return rho_out.value, count.value

    def close(self):
        if self.handle:
            self.lib.trlda_model_destroy(self.handle)
            self.handle = None

    def set_alpha(self, alpha):
        _ffi.check(self.lib.trlda_model_set_alpha(self.handle, np.asfortranarray(alpha)))

    def set_lambda(self, lam):
        _ffi.check(self.lib.trlda_model_set_lambda(self.handle, np.asfortranarray(lam)))

    def get_lambda(self):
        lam = np.empty((self.K, self.V), dtype=np.float64, order="F")
        _ffi.check(self.lib.trlda_model_get_lambda(self.handle, lam))
        return lam

    def upload(self, csr):
        return DeviceBatch(csr, self.V, self.device_index)

    def snapshot_lambda(self):
        _ffi.check(self.lib.trlda_model_copy_lambda(self.handle, self.lambda_prime.data_ptr()))

    def wordcounts(self, batch):
        _ffi.check(self.lib.trlda_model_wordcounts(self.handle, batch.handle, self.wc.data_ptr()))
        return self.wc

    def tr_init(self, wc, rho, eta, coef):
        _ffi.check(self.lib.trlda_model_tr_init_wc(self.handle, wc.data_ptr(),
                                                   self.lambda_prime.data_ptr(), rho, eta, coef))

    def estep(self, batch, gamma0, max_iter, threshold):
        """gamma0: host K x B_local array, or None to warm-start from the last call."""
        torch = self.torch
        if gamma0 is not None:
            host = torch.from_numpy(np.ascontiguousarray(np.asfortranarray(gamma0).T))
            self.gamma = host.to(self.device)          # B x K C-order == K x B column-major
        _ffi.check(self.lib.trlda_model_estep(self.handle, batch.handle, self.gamma.data_ptr(),
                                              self.sstats.data_ptr(), int(max_iter),
                                              float(threshold), None))
        return self.sstats

    def blend(self, sstats, rho, eta, scale):
        _ffi.check(self.lib.trlda_model_blend(self.handle, self.lambda_prime.data_ptr(),
                                              sstats.data_ptr(), rho, eta, scale))

    def gamma_host(self):
        return np.asfortranarray(self.gamma.cpu().numpy().T)

    def draw_gamma(self, total_docs, lo, hi):
        """gamma0 = columns [lo, hi) of sampleGamma(K, total_docs, 100) / 100 (lda.cpp:135), drawn
        on the device from the host's libc stream (csrc/rng_kernels.h); the stream advances by
        the whole matrix, so every rank stays in step."""
        self.gamma = self.torch.empty(max(hi - lo, 1) * self.K, dtype=self.torch.float64,
                                      device=self.device)
        _ffi.check(self.lib.trlda_model_sample_gamma_cols(self.handle, self.K, int(total_docs),
                                                          int(lo), int(hi), 100, 100.,
                                                          self.gamma.data_ptr()))


class ShardedOnlineLDA(object):
    """OnlineLDA whose ``update_parameters`` runs data-parallel over a process group.

    Every rank calls ``update_parameters`` with the SAME full mini-batch (or, with
    ``presharded=True``, with its own shard); results equal the single-GPU model's up
    to the summation order of the all-reduce.
    """

    def __init__(self, num_words, num_topics, num_documents, alpha=.1, eta=.3, group=None,
                 engine=None, device=None, gamma_init="replicated", exchange="auto"):
        import torch.distributed as dist
        from .models import _alpha_vector, _default_device
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        K, alpha_vec = _alpha_vector(alpha, num_topics)
        self._K, self._V = K, int(num_words)
        self._alpha, self._eta = alpha_vec, float(eta)
        self.num_documents = int(num_documents)
        self.update_count = 0
        if gamma_init not in ("replicated", "local"):
            raise ValueError("gamma_init must be 'replicated' or 'local'")
        self.gamma_init = gamma_init
        if exchange not in ("auto", "factors", "sstats"):
            raise ValueError("exchange must be 'auto', 'factors' or 'sstats'")
        self.exchange = exchange
        if engine is None:
            _ffi.require_gpu()
            engine = HipEngine(self._V, K, _default_device() if device is None else device)
        self.engine = engine
        engine.set_alpha(alpha_vec)
        # lambda = sampleGamma(K, V, 100) / 100 (lda.cpp:71), drawn on every rank (so that a
        # rank's stream is where the single-process run's would be) -- and then rank 0's lambda
        # AND rank 0's generator state are broadcast: the library seeds itself from the clock at
        # load time (module.cpp:356-359), so without `trlda_amd.seed(s)` on every rank the
        # replicas would start from different lambdas and draw different "replicated" gamma0s,
        # and never agree.
        lam = np.empty((K, self._V), dtype=np.float64, order="F")
        _ffi.lib().trlda_sample_gamma_init(K, self._V, lam)
        if self.world > 1:
            lam = self._broadcast_host(lam)
            state = np.zeros(33, dtype=np.uint32)
            _ffi.lib().trlda_rng_get_state(state)
            state = self._broadcast_host(state)
            _ffi.lib().trlda_rng_set_state(state)
        engine.set_lambda(lam)
        # the factor exchange runs ncclAllGather on a communicator of our own
        self._factors_ok = hasattr(engine, "update_dp") and gamma_init == "replicated"
        if self._factors_ok and self.world > 1 and exchange != "sstats":
            self._factors_ok = dist.get_backend(group) == "nccl" and \
                engine.make_communicator(dist, group) is not None
            self._factors_ok = self._all_ranks(self._factors_ok)

    def _all_ranks(self, flag):
        """True when `flag` holds on every rank"""
        import torch
        t = torch.tensor([int(bool(flag))])
        if self.dist.get_backend(self.group) == "nccl":
            t = t.to(self.engine.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def use_factors(self, csr, cuts):
        """The factor exchange moves world * slot doubles per E-step, slot = max_r(docs_r) * K +
        max_r(nnz_r); the all-reduce about 2 * K * V (reduce-scatter + all-gather)."""
        if not self._factors_ok or self.exchange == "sstats":
            return False
        if self.exchange == "factors":
            return True
        docs = int(np.max(np.diff(cuts))) if len(cuts) > 1 else 0
        nnz = int(np.max(np.diff(csr.indptr[cuts]))) if len(cuts) > 1 else 0
        return self.world * (docs * self._K + nnz) < 2 * self._K * self._V

    num_topics = property(lambda self: self._K)
    num_words = property(lambda self: self._V)
    eta = property(lambda self: self._eta)

    @property
    def lambdas(self):
        lam = self.engine.get_lambda()
        lam.flags.writeable = False
        return lam

    @lambdas.setter
    def lambdas(self, value):
        arr = np.asarray(value, dtype=np.float64)
        if arr.shape != (self._K, self._V):
            raise RuntimeError("Lambda has wrong dimensionality.")
        self.engine.set_lambda(arr)

    # -- helpers -----------------------------------------------------------------------
    def _broadcast_host(self, array):
        """Rank 0's copy of a host array on every rank (through the group's backend: RCCL moves
        device memory, gloo host memory)."""
        import torch
        flat = np.ascontiguousarray(array).reshape(-1)
        signed = flat.view(np.int32) if flat.dtype == np.uint32 else flat
        t = torch.from_numpy(signed.copy())
        backend = self.dist.get_backend(self.group)
        if backend == "nccl":
            t = t.to(self.engine.device)
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, 0) if self.group else 0,
                            group=self.group)
        out = t.cpu().numpy()
        if flat.dtype == np.uint32:
            out = out.view(np.uint32)
        out = out.reshape(array.shape)
        return np.asfortranarray(out) if array.ndim == 2 else out

    def replicas_agree(self):
        """True when every rank holds the same lambda (a checksum of it is compared across the
        group): what the replicated M-step relies on."""
        import torch
        lam = self.engine.get_lambda()
        digest = np.array([float(lam.sum()), float(np.abs(lam).max()), float(lam[:, ::7].sum())])
        if self.world == 1:
            return True
        lo, hi = torch.from_numpy(digest.copy()), torch.from_numpy(digest.copy())
        if self.dist.get_backend(self.group) == "nccl":
            lo, hi = lo.to(self.engine.device), hi.to(self.engine.device)
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.group)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(torch.equal(lo.cpu(), hi.cpu()))

    def _all_reduce(self, tensor):
        if self.world > 1:
            self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group)
        return tensor

    def _fresh_gamma(self, total_docs, lo, hi):
        """gamma0 for documents [lo, hi) of a total_docs-document mini-batch."""
        L = _ffi.lib()
        if self.gamma_init == "replicated":
            # the exact stream of the single-process run: draw the whole K x B matrix
            # (lda.cpp:135) on every rank and keep this rank's columns
            full = np.empty((self._K, total_docs), dtype=np.float64, order="F")
            L.trlda_sample_gamma_init(self._K, total_docs, full)
            return np.asfortranarray(full[:, lo:hi])
        local = np.empty((self._K, hi - lo), dtype=np.float64, order="F")
        L.trlda_sample_gamma_init(self._K, hi - lo, local)
        return local

    # -- the update ----------------------------------------------------------------------
    def update_parameters(self, docs, max_iter_tr=10, max_iter_inference=20, kappa=.7, tau=100.,
                          rho=-1., init_gamma=True, update_lambda=True, presharded=False,
                          total_docs=None, doc_range=None, threshold=0.001):
        """onlinelda.cpp:53-111,177-179 with the E-step sharded over ranks; returns rho."""
        csr = as_csr(docs)
        if presharded:
            if total_docs is None or doc_range is None:
                raise ValueError("presharded=True needs total_docs and doc_range=(lo, hi)")
            shard, (lo, hi), B = csr, doc_range, int(total_docs)
        else:
            B = len(csr)
            cuts = csr.shard_cuts(self.world)
            lo, hi = int(cuts[self.rank]), int(cuts[self.rank + 1])
            shard = csr.slice(lo, hi)
        if B == 0:
            return 1.0                                           # onlinelda.cpp:54-56
        if rho < 0.:
            rho = float(np.power(tau + self.update_count, -kappa))   # onlinelda.cpp:59-66
        if update_lambda and not presharded and self.use_factors(csr, cuts):
            # every rank holds the whole mini-batch: documents of this rank -> all-gather of the
            # factors -> statistics of the whole mini-batch + fused M-step on every rank
            eng = self.engine
            whole, mine = eng.upload(csr), eng.upload(shard)
            try:
                rho, self.update_count = eng.update_dp(
                    whole, mine, cuts, self.rank, self.world, self.num_documents, self._eta,
                    max_iter_tr, max_iter_inference, kappa, tau, rho, init_gamma, threshold,
                    self.update_count)
            finally:
                whole.close()
                mine.close()
            return rho
        if update_lambda:
            eng = self.engine
            batch = eng.upload(shard)
            try:
                eng.snapshot_lambda()                            # lambdaPrime = mLambda
                scale = float(self.num_documents) / float(B)
                n_steps = max_iter_tr if max_iter_tr > 0 else 1
                if max_iter_tr > 0:
                    wc = self._all_reduce(eng.wordcounts(batch))
                    coef = float(self.num_documents) / float(B) / float(self._K)
                    eng.tr_init(wc, rho, self._eta, coef)
                for i in range(n_steps):
                    fresh = not (i > 0 and init_gamma)           # onlinelda.cpp:91-95
                    g0 = None
                    if fresh and hasattr(eng, "draw_gamma"):     # on the device, same stream
                        if self.gamma_init == "replicated":
                            eng.draw_gamma(B, lo, hi)
                        else:
                            eng.draw_gamma(hi - lo, 0, hi - lo)
                    elif fresh:
                        g0 = self._fresh_gamma(B, lo, hi)
                    sstats = eng.estep(batch, g0, max_iter_inference, threshold)
                    sstats = self._all_reduce(sstats)
                    eng.blend(sstats, rho, self._eta, scale)     # onlinelda.cpp:99-100
            finally:
                if hasattr(batch, "close"):
                    batch.close()
        self.update_count += 1                                   # onlinelda.cpp:177
        return rho
