"""Data-parallel OnlineLDA and BatchLDA over the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm).

Documents are independent given lambda (reference src/lda.cpp:176-214 touches only column i of
gamma and *adds* into sstats), so a mini-batch is split into contiguous document ranges, one per
rank, balanced by nnz.  lambda is replicated; the ranks meet where the reference has its
``omp critical`` reduction (lda.cpp:211-217), in one of two ways chosen per call by the bytes
they move (``exchange="auto"``):

* the all-reduce of the K x V statistics: every rank computes its own preamble and its shard's
  statistics, ONE all-reduce sums them, every rank applies the identical M-step
  (onlinelda.cpp:99-100 / batchlda.cpp:60), so lambda stays replicated without a broadcast;
  the trust-region initial step (onlinelda.cpp:79-86) needs one more all-reduce of V word
  counts per call.  With a RCCL communicator of the process's own (``rccl.py``) the whole call is
  one C-ABI entry (``trlda_model_online_update_multi`` / ``trlda_model_batch_update_multi``);
  without one (gloo on CPU, a failed ncclCommInitRank) the same steps are composed here from
  single calls and ``torch.distributed.all_reduce``;
* the FACTOR exchange of csrc/dp_kernels.h: an all-gather of every document's expElogtheta row
  and per-entry weights (8 (K + n_d) bytes per document), after which every rank forms the whole
  mini-batch's statistics and the fused M-step itself (``trlda_model_online_update_dp`` /
  ``trlda_model_batch_update_dp``).

The empirical-Bayes steps (onlinelda.cpp:116-162, batchlda.cpp:64-205) need one K-vector summed
over the ranks' documents (``trlda_model_eb_gamma_stats_multi``, onlinelda.cpp:128 across ranks);
everything else they read -- lambda, its row sums -- is replicated, and so are the K-sized Newton
steps.  The adaptive learning rate (onlinelda.cpp:167-175) reads lambda' and the reduced
statistics, both replicated.

The per-rank computation is behind a small *engine* interface so that the sharding / collective
/ learning-rate logic can be exercised on CPU with the gloo backend
(tests/test_distributed_cpu.py plugs in a checker-backed engine there); the product engine is
:class:`HipEngine` and nothing else ships.
"""
import ctypes as C
import math

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
        self._lambda_prime = self._sstats = self._wc = None   # composition path only: on demand
        self.gamma = None             # composition path: this rank's K x B_local, (B_local, K)
        self.gamma_internal = False   # the last E-step left gamma in the model's own workspace
        self.comm = None

    # -- the composition path's K x V buffers, allocated when that path first runs ----------
    @property
    def lambda_prime(self):
        if self._lambda_prime is None:
            self._lambda_prime = self.torch.empty(self.K * self.V, dtype=self.torch.float64,
                                                  device=self.device)
        return self._lambda_prime

    @property
    def sstats(self):
        if self._sstats is None:
            self._sstats = self.torch.empty(self.K * self.V, dtype=self.torch.float64,
                                            device=self.device)
        return self._sstats

    @property
    def wc(self):
        if self._wc is None:
            self._wc = self.torch.empty(self.V, dtype=self.torch.float64, device=self.device)
        return self._wc

    def make_communicator(self, dist, group):
        """A ncclComm_t of our own over the group's ranks (collective call); None if RCCL cannot
        be reached that way."""
        from . import rccl
        try:
            self.comm = rccl.own_communicator(dist, self.device, group)
        except Exception:                             # noqa: BLE001 -- the all-reduce path remains
            self.comm = None
        return self.comm

    def connect_direct(self, dist, group, rank, world, max_slot):
        """The direct slot exchange (include/trlda_hip.h, trlda_model_dp_direct_*): export this
        model's region, hand the 64-byte handle to every peer through the group, map theirs.
        Collective; returns False (on every rank alike) if any rank could not."""
        torch = self.torch
        mine = C.create_string_buffer(64)
        ok = self.lib.trlda_model_dp_direct_alloc(self.handle, int(max_slot), int(world), mine) == 0
        backend_dev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.frombuffer(bytearray(mine.raw + bytes([int(ok)])), dtype=torch.uint8).clone().to(backend_dev)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        raw = [p.cpu().numpy().tobytes() for p in parts]
        ok = all(r[64] == 1 for r in raw)
        if ok:
            handles = b"".join(r[:64] for r in raw)
            ok = self.lib.trlda_model_dp_direct_connect(self.handle, int(rank), int(world), handles) == 0
        flag = torch.tensor([int(ok)]).to(backend_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            self.lib.trlda_model_dp_direct_close(self.handle)
            return False
        return True

    def drop_communicator(self):
        from . import rccl
        if self.comm is not None:
            rccl.destroy(self.comm)
            self.comm = None

    def close(self):
        if self.handle:
            self.lib.trlda_model_synchronize(self.handle)
            self.drop_communicator()                  # before the model: nothing of it in flight
            self.lib.trlda_model_destroy(self.handle)
            self.handle = None

    # -- whole calls through the C ABI (need the communicator, or world 1) --------------------
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
        self.gamma_internal = True
        return rho_out.value, count.value

    def update_multi(self, shard, total_docs, doc_lo, num_documents, eta, max_iter_tr,
                     max_iter_inference, kappa, tau, rho, init_gamma, threshold, update_count):
        """onlinelda.cpp:53-111 with the all-reduce of the statistics, one C call."""
        count, rho_out = C.c_int(int(update_count)), C.c_double(0.)
        _ffi.check(self.lib.trlda_model_online_update_multi(
            self.handle, shard.handle, self.comm, int(total_docs), int(doc_lo), int(num_documents),
            float(eta), int(max_iter_tr), int(max_iter_inference), float(kappa), float(tau),
            float(rho), int(bool(init_gamma)), float(threshold), C.byref(count),
            C.byref(rho_out)))
        self.gamma_internal = True
        return rho_out.value, count.value

    def batch_update_dp(self, batch, shard, cuts, rank, world, eta, max_epochs,
                        max_iter_inference, threshold):
        """batchlda.cpp:43-61 with the factor exchange."""
        cuts = np.ascontiguousarray(cuts, dtype=np.int32)
        _ffi.check(self.lib.trlda_model_batch_update_dp(
            self.handle, batch.handle, shard.handle, self.comm, int(rank), int(world),
            cuts.ctypes.data_as(C.POINTER(C.c_int32)), float(eta), int(max_epochs),
            int(max_iter_inference), 1, float(threshold)))
        self.gamma_internal = True

    def batch_update_multi(self, shard, total_docs, doc_lo, eta, max_epochs, max_iter_inference,
                           threshold):
        """batchlda.cpp:43-61 with the all-reduce of the statistics, one C call."""
        _ffi.check(self.lib.trlda_model_batch_update_multi(
            self.handle, shard.handle, self.comm, int(total_docs), int(doc_lo), float(eta),
            int(max_epochs), int(max_iter_inference), 1, float(threshold)))
        self.gamma_internal = True

    # -- single steps (the composition path, and what both paths share) ----------------------
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

    def set_keep_sstats(self, keep):
        _ffi.check(self.lib.trlda_model_set_keep_sstats(self.handle, int(bool(keep))))

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
        self.gamma_internal = False
        return self.sstats

    def blend(self, sstats, rho, eta, scale):
        _ffi.check(self.lib.trlda_model_blend(self.handle, self.lambda_prime.data_ptr(),
                                              sstats.data_ptr(), rho, eta, scale))

    def gamma_host(self):
        """This rank's gamma of the composition path's last E-step, K x B_local."""
        return np.asfortranarray(self.gamma.cpu().numpy().T)

    def draw_gamma(self, total_docs, lo, hi):
        """gamma0 = columns [lo, hi) of sampleGamma(K, total_docs, 100) / 100 (lda.cpp:135), drawn
        on the device from the host's libc stream (csrc/rng_kernels.h); the stream advances by
        the whole matrix, so every rank stays in step."""
        # (hi - lo, K) C-order == K x (hi - lo) column-major; an empty shard keeps a 0 x K tensor
        # (the draw still has to run: it moves the stream)
        self.gamma = self.torch.empty((hi - lo, self.K), dtype=self.torch.float64,
                                      device=self.device)
        scratch = self.gamma if hi > lo else self.torch.empty((1, self.K), dtype=self.torch.float64,
                                                              device=self.device)
        _ffi.check(self.lib.trlda_model_sample_gamma_cols(self.handle, self.K, int(total_docs),
                                                          int(lo), int(hi), 100, 100.,
                                                          scratch.data_ptr()))

    def resident_estep(self, shard, total_docs, doc_lo, max_iter, threshold):
        """updateVariables from a fresh gamma for this rank's documents, gamma left on the device
        (onlinelda.cpp:118-120 when update_lambda is off)."""
        _ffi.check(self.lib.trlda_model_estep_resident_shard(
            self.handle, shard.handle, int(total_docs), int(doc_lo), int(max_iter),
            float(threshold)))
        self.gamma_internal = True

    def psi_gamma_diff(self, num_local):
        """(sum over documents of psi(gamma_dk) - psi(sum_k gamma_dk), already_summed_over_ranks):
        over ALL ranks' documents when this engine has a communicator (one all-reduce of K
        doubles on the model's stream), else this rank's share."""
        out = np.zeros(self.K, dtype=np.float64)
        gamma = None if self.gamma_internal else \
            (C.c_void_p(self.gamma.data_ptr()) if num_local > 0 else None)
        _ffi.check(self.lib.trlda_model_eb_gamma_stats_multi(self.handle, self.comm, int(num_local),
                                                             gamma, out))
        return out, self.comm is not None

    def lambda_psi_stats(self):
        """(sum_kw psi(lambda_kw), row sums of lambda): replicated, no exchange."""
        total = C.c_double(0.)
        rowsums = np.empty(self.K, dtype=np.float64)
        _ffi.check(self.lib.trlda_model_eb_lambda_stats(self.handle, C.byref(total), rowsums))
        return total.value, rowsums

    def adaptive_stats(self, eta, scale, tau, internal):
        """(|lambdaUpdate|^2, |mAdaGradient|^2) of onlinelda.cpp:167-172 from the reduced
        statistics and lambda': the model's own (C paths) or this engine's tensors."""
        u2, g2 = C.c_double(0.), C.c_double(0.)
        if internal:
            _ffi.check(self.lib.trlda_model_adaptive_stats(self.handle, float(eta), float(scale),
                                                           float(tau), C.byref(u2), C.byref(g2)))
        else:
            _ffi.check(self.lib.trlda_model_adaptive_stats_dev(
                self.handle, self.sstats.data_ptr(), self.lambda_prime.data_ptr(), float(eta),
                float(scale), float(tau), C.byref(u2), C.byref(g2)))
        return u2.value, g2.value


class _ShardedLDA(object):
    """What the sharded models share: the group, the replicated state and its consistency, the
    cut of a mini-batch into per-rank document ranges, the choice of the exchange."""

    def _setup(self, num_words, num_topics, alpha, eta, group, engine, device, gamma_init,
               exchange, own_communicator="auto", direct_exchange=False):
        import torch.distributed as dist
        from .models import _alpha_vector, _default_device
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        K, alpha_vec = _alpha_vector(alpha, num_topics)
        self._K, self._V = K, int(num_words)
        self._alpha, self._eta = alpha_vec, float(eta)
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
        self.last_path = None         # how the last update met the other ranks (tests, bench)
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
        # a RCCL communicator of our own: the whole-call C entry points and the factor exchange
        # run their collectives on it (torch keeps its communicators private)
        # (own_communicator=True: also for a group of one rank -- the same C entry points with a
        # real ncclComm_t, which is how a one-GPU box exercises them)
        self._own_comm = False
        if (self.world > 1 or own_communicator is True) and dist.is_initialized() and \
                hasattr(engine, "make_communicator") and dist.get_backend(group) == "nccl":
            self._own_comm = self._all_ranks(engine.make_communicator(dist, group) is not None)
            if not self._own_comm:
                engine.drop_communicator()            # all ranks use it, or none does
        # direct_exchange: the factor slots written straight into the peers' buffers (hipIpc)
        # instead of an all-gather; the region is made (and every rank's handle exchanged) when
        # the first mini-batch says how large a slot is, again when one needs more
        self._direct = bool(direct_exchange) and self.world > 1 and hasattr(engine, "connect_direct")
        self._direct_slot = 0
        self._factors_ok = hasattr(engine, "update_dp") and gamma_init == "replicated" and \
            exchange != "sstats" and (self.world == 1 or self._own_comm or self._direct)

    # -- replicated state -----------------------------------------------------------------
    num_topics = property(lambda self: self._K)
    num_words = property(lambda self: self._V)

    @property
    def eta(self):
        return self._eta

    @property
    def alpha(self):
        return self._alpha.reshape(-1, 1).copy(order="F")

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

    def close(self):
        if hasattr(self.engine, "close"):
            self.engine.close()

    # -- helpers -----------------------------------------------------------------------
    def _on_group_device(self, tensor):
        if self.dist.get_backend(self.group) == "nccl":
            return tensor.to(self.engine.device)
        return tensor

    def _all_ranks(self, flag):
        """True when `flag` holds on every rank"""
        import torch
        t = self._on_group_device(torch.tensor([int(bool(flag))]))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def _broadcast_host(self, array):
        """Rank 0's copy of a host array on every rank (through the group's backend: RCCL moves
        device memory, gloo host memory)."""
        import torch
        flat = np.ascontiguousarray(array).reshape(-1)
        signed = flat.view(np.int32) if flat.dtype == np.uint32 else flat
        t = self._on_group_device(torch.from_numpy(signed.copy()))
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, 0) if self.group else 0,
                            group=self.group)
        out = t.cpu().numpy()
        if flat.dtype == np.uint32:
            out = out.view(np.uint32)
        out = out.reshape(array.shape)
        return np.asfortranarray(out) if array.ndim == 2 else out

    def _sum_host(self, array):
        """Sum over ranks of a small host array (K doubles: onlinelda.cpp:128 across ranks)."""
        import torch
        if self.world == 1:
            return array
        t = self._on_group_device(torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def replicas_agree(self):
        """True when every rank holds the same lambda, alpha and eta (checksums compared across
        the group): what the replicated M-step and Newton steps rely on."""
        import torch
        lam = self.engine.get_lambda()
        digest = np.array([float(lam.sum()), float(np.abs(lam).max()), float(lam[:, ::7].sum()),
                           float(self._alpha.sum()), float(self._eta)])
        if self.world == 1:
            return True
        lo = self._on_group_device(torch.from_numpy(digest.copy()))
        hi = self._on_group_device(torch.from_numpy(digest.copy()))
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.group)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(torch.equal(lo.cpu(), hi.cpu()))

    def _all_reduce(self, tensor):
        if self.world > 1:
            self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group)
        return tensor

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

    def _ensure_direct(self, csr, cuts):
        """The direct exchange region covers this mini-batch's slots (collective when it grows)."""
        if not self._direct:
            return
        docs = int(np.max(np.diff(cuts)))
        nnz = int(np.max(np.diff(csr.indptr[cuts])))
        need = (docs * self._K + nnz + self._K - 1) // self._K * self._K
        if need <= self._direct_slot:
            return
        want = max(2 * need, 1 << 16)
        if self.engine.connect_direct(self.dist, self.group, self.rank, self.world, want):
            self._direct_slot = want
        else:                                        # every rank falls back together
            self._direct, self._direct_slot = False, 0
            self._factors_ok = self._factors_ok and self._own_comm

    def _cut(self, docs, presharded, total_docs, doc_range):
        """-> (whole mini-batch or None, this rank's shard, cuts or None, lo, hi, B)"""
        csr = as_csr(docs)
        if presharded:
            if total_docs is None or doc_range is None:
                raise ValueError("presharded=True needs total_docs and doc_range=(lo, hi)")
            lo, hi = int(doc_range[0]), int(doc_range[1])
            if hi - lo != len(csr):
                raise ValueError("doc_range does not match the number of documents handed in")
            return None, csr, None, lo, hi, int(total_docs)
        cuts = csr.shard_cuts(self.world)
        lo, hi = int(cuts[self.rank]), int(cuts[self.rank + 1])
        return csr, csr.slice(lo, hi), cuts, lo, hi, len(csr)

    def _fresh_gamma(self, total_docs, lo, hi):
        """gamma0 for documents [lo, hi) of a total_docs-document mini-batch, on the host."""
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

    def _composed_estep(self, batch, fresh, B, lo, hi, max_iter, threshold):
        """One E-step of the composition path on this rank's shard; returns the shard's
        statistics (a tensor the caller all-reduces)."""
        eng = self.engine
        g0 = None
        if fresh and hasattr(eng, "draw_gamma"):             # on the device, same stream
            if self.gamma_init == "replicated":
                eng.draw_gamma(B, lo, hi)
            else:
                eng.draw_gamma(hi - lo, 0, hi - lo)
        elif fresh:
            g0 = self._fresh_gamma(B, lo, hi)
        return eng.estep(batch, g0, max_iter, threshold)

    def _psi_gamma_diff(self, num_local):
        """sum over the WHOLE mini-batch's documents of psi(gamma_dk) - psi(sum_k gamma_dk)."""
        vec, summed = self.engine.psi_gamma_diff(num_local)
        return vec if summed else self._sum_host(vec)

    def _c_calls(self):
        """The whole-call C entry points (all-reduce on a communicator of our own) can run."""
        return hasattr(self.engine, "update_multi") and self.gamma_init == "replicated" and \
            self._own_comm


class ShardedOnlineLDA(_ShardedLDA):
    """OnlineLDA whose ``update_parameters`` runs data-parallel over a process group.

    Every rank calls ``update_parameters`` with the SAME full mini-batch (or, with
    ``presharded=True``, with its own shard); results equal the single-GPU model's up
    to the summation order of the all-reduce.
    """

    def __init__(self, num_words, num_topics, num_documents, alpha=.1, eta=.3, group=None,
                 engine=None, device=None, gamma_init="replicated", exchange="auto",
                 own_communicator="auto", direct_exchange=False):
        self.num_documents = int(num_documents)
        self.update_count = 0
        # adaptive learning rate state (onlinelda.cpp:28-31)
        self._ada_tau, self._ada_rho, self._ada_sq_norm = 1000., 1. / 1000., 1.
        self._setup(num_words, num_topics, alpha, eta, group, engine, device, gamma_init, exchange,
                    own_communicator, direct_exchange)

    def update_parameters(self, docs, max_iter_tr=10, max_iter_inference=20, kappa=.7, tau=100.,
                          rho=-1., adaptive=False, init_gamma=True, update_lambda=True,
                          update_alpha=False, update_eta=False, min_alpha=1e-6, min_eta=1e-6,
                          verbosity=0, presharded=False, total_docs=None, doc_range=None,
                          threshold=0.001):
        """onlinelda.cpp:53-179 with the E-step sharded over ranks; returns rho."""
        from .models import _online_alpha_step, _online_eta_step
        csr, shard, cuts, lo, hi, B = self._cut(docs, presharded, total_docs, doc_range)
        if B == 0:
            return 1.0                                           # onlinelda.cpp:54-56
        rho = float(rho)
        if rho < 0. and adaptive:
            rho = self._ada_rho                                  # onlinelda.cpp:61-62
        if rho < 0.:
            rho = math.pow(tau + self.update_count, -kappa)   # libm, as onlinelda.cpp:59-66 (numpy.power differs by an ulp at e.g. (12, -.9))
        eng = self.engine
        eta_old = self._eta
        keep = bool(adaptive and update_lambda)
        if hasattr(eng, "set_keep_sstats"):
            eng.set_keep_sstats(keep)
        internal = False             # where the reduced statistics / lambda' of this call lie
        whole = mine = None
        try:
            mine = eng.upload(shard)
            factors = update_lambda and csr is not None and self.use_factors(csr, cuts)
            if factors:
                self._ensure_direct(csr, cuts)       # (may fall back, on every rank alike)
                factors = self.use_factors(csr, cuts)
            if factors:
                # every rank holds the whole mini-batch: documents of this rank -> all-gather of
                # the factors -> statistics of the whole mini-batch + fused M-step on every rank
                whole = eng.upload(csr)
                self.last_path = "factors-direct" if self._direct else "factors"
                rho, self.update_count = eng.update_dp(
                    whole, mine, cuts, self.rank, self.world, self.num_documents, self._eta,
                    max_iter_tr, max_iter_inference, kappa, tau, rho, init_gamma, threshold,
                    self.update_count)
                internal = True
            elif update_lambda and self._c_calls():
                self.last_path = "allreduce"
                rho, self.update_count = eng.update_multi(
                    mine, B, lo, self.num_documents, self._eta, max_iter_tr, max_iter_inference,
                    kappa, tau, rho, init_gamma, threshold, self.update_count)
                internal = True
            elif update_lambda:
                self.last_path = "allreduce-composed"
                eng.snapshot_lambda()                            # lambdaPrime = mLambda
                scale = float(self.num_documents) / float(B)
                n_steps = max_iter_tr if max_iter_tr > 0 else 1
                if max_iter_tr > 0:
                    wc = self._all_reduce(eng.wordcounts(mine))
                    coef = float(self.num_documents) / float(B) / float(self._K)
                    eng.tr_init(wc, rho, self._eta, coef)
                for i in range(n_steps):
                    fresh = not (i > 0 and init_gamma)           # onlinelda.cpp:91-95
                    sstats = self._composed_estep(mine, fresh, B, lo, hi, max_iter_inference,
                                                  threshold)
                    sstats = self._all_reduce(sstats)
                    eng.blend(sstats, rho, self._eta, scale)     # onlinelda.cpp:99-100
                self.update_count += 1                           # onlinelda.cpp:177
            else:
                self.update_count += 1

            if update_alpha:                                     # onlinelda.cpp:116-142
                if not update_lambda:
                    if hasattr(eng, "resident_estep") and self.gamma_init == "replicated":
                        eng.resident_estep(mine, B, lo, max_iter_inference, threshold)
                    else:
                        self._composed_estep(mine, True, B, lo, hi, max_iter_inference, threshold)
                alpha = _online_alpha_step(self._alpha, self._psi_gamma_diff(hi - lo), B, rho,
                                           min_alpha)
                eng.set_alpha(alpha)
                self._alpha = alpha
            if update_eta:                                       # onlinelda.cpp:147-162
                sum_psi, rowsums = eng.lambda_psi_stats()
                self._eta = _online_eta_step(self._eta, sum_psi, rowsums, self._K, self._V, rho,
                                             min_eta)
            if keep:                                             # onlinelda.cpp:167-175
                t = self._ada_tau
                u2, g2 = eng.adaptive_stats(eta_old, float(self.num_documents) / B, t, internal)
                self._ada_sq_norm = (1. - 1. / t) * self._ada_sq_norm + 1. / t * u2
                self._ada_rho = g2 / self._ada_sq_norm
                self._ada_tau = t * (1. - self._ada_rho) + 1.
        finally:
            for b in (whole, mine):
                if b is not None and hasattr(b, "close"):
                    b.close()
        return rho


class ShardedBatchLDA(_ShardedLDA):
    """BatchLDA (src/batchlda.cpp) whose ``update_parameters`` runs data-parallel over a process
    group: BASELINE.json's configuration 4 (K = 200, V = 50 000, 100 000 documents over 8 GPUs).

    Every rank calls ``update_parameters`` with the SAME documents (or, with ``presharded=True``,
    with its own contiguous range of them).  Per epoch: every rank's documents from a fresh
    gamma, one all-reduce of the K x V statistics (80 MB at config 4) or one all-gather of the
    documents' factors, whichever moves fewer bytes, ``lambda = eta + sstats`` on every rank.
    """

    def __init__(self, num_words, num_topics, alpha=.1, eta=.3, group=None, engine=None,
                 device=None, gamma_init="replicated", exchange="auto", own_communicator="auto",
                 direct_exchange=False):
        self._setup(num_words, num_topics, alpha, eta, group, engine, device, gamma_init, exchange,
                    own_communicator, direct_exchange)

    def update_parameters(self, docs, max_epochs=100, max_iter_inference=100, max_iter_alpha=10,
                          max_iter_eta=20, update_lambda=True, update_alpha=False,
                          update_eta=False, min_alpha=1e-6, min_eta=1e-6,
                          emp_bayes_threshold=1e-8, verbosity=0, presharded=False,
                          total_docs=None, doc_range=None, threshold=0.001):
        """batchlda.cpp:43-208 with the E-steps sharded over ranks; returns 1."""
        from .models import _alpha_line_search, _eta_line_search
        csr, shard, cuts, lo, hi, B = self._cut(docs, presharded, total_docs, doc_range)
        if B == 0:
            return 1.                                            # batchlda.cpp:44-46
        eng = self.engine
        if hasattr(eng, "set_keep_sstats"):
            eng.set_keep_sstats(False)
        emp_bayes = update_alpha or update_eta
        factors = update_lambda and csr is not None and self.use_factors(csr, cuts)
        if factors:
            self._ensure_direct(csr, cuts)           # (may fall back, on every rank alike)
            factors = self.use_factors(csr, cuts)
        whole = mine = None
        try:
            mine = eng.upload(shard)
            if factors:
                whole = eng.upload(csr)

            def lambda_epochs(n):                                # batchlda.cpp:48-61
                self.last_path = ("factors-direct" if self._direct else "factors") if factors else \
                    "allreduce" if self._c_calls() else "allreduce-composed"
                if factors:
                    eng.batch_update_dp(whole, mine, cuts, self.rank, self.world, self._eta, n,
                                        max_iter_inference, threshold)
                elif self._c_calls():
                    eng.batch_update_multi(mine, B, lo, self._eta, n, max_iter_inference, threshold)
                else:
                    for _ in range(n):
                        sstats = self._composed_estep(mine, True, B, lo, hi, max_iter_inference,
                                                      threshold)
                        sstats = self._all_reduce(sstats)
                        eng.snapshot_lambda()        # any finite lambda': it is multiplied by 0
                        eng.blend(sstats, 1., self._eta, 1.)     # lambda = eta + sstats   (:60)

            if not emp_bayes:
                if update_lambda:
                    lambda_epochs(int(max_epochs))
                return 1.
            for _epoch in range(int(max_epochs)):
                if update_lambda:
                    lambda_epochs(1)
                if update_alpha:                                 # batchlda.cpp:64-142
                    if not update_lambda:
                        if hasattr(eng, "resident_estep") and self.gamma_init == "replicated":
                            eng.resident_estep(mine, B, lo, max_iter_inference, threshold)
                        else:
                            self._composed_estep(mine, True, B, lo, hi, max_iter_inference,
                                                 threshold)
                    alpha = _alpha_line_search(self._alpha, self._psi_gamma_diff(hi - lo), B,
                                               max_iter_alpha, min_alpha, emp_bayes_threshold,
                                               verbosity if self.rank == 0 else 0)
                    eng.set_alpha(alpha)
                    self._alpha = alpha
                if update_eta:                                   # batchlda.cpp:147-205
                    sum_psi, rowsums = eng.lambda_psi_stats()
                    self._eta = _eta_line_search(self._eta, sum_psi, rowsums, self._K, self._V,
                                                 max_iter_eta, min_eta, emp_bayes_threshold,
                                                 verbosity if self.rank == 0 else 0)
        finally:
            for b in (whole, mine):
                if b is not None and hasattr(b, "close"):
                    b.close()
        return 1.                                                # batchlda.cpp:207
