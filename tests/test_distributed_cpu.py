"""The N>1 path on CPU: world_size-2 gloo process group.

What ships for multi-GPU is host logic in trlda_amd/distributed.py (shard by nnz, local
E-step, ONE all-reduce of the K x V statistics, replicated M-step, learning-rate / counter
bookkeeping) around the per-rank HIP engine.  Here the engine is replaced by a
checker-backed stand-in (the CPU oracle -- test infrastructure, this file only) so that the
sharding and the collectives run for real over gloo, and the result is compared with the
single-process trajectory.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine(object):
    """Engine interface of trlda_amd.distributed backed by oracle/cpu_ref.c, CPU tensors."""

    def __init__(self, V, K):
        from oracle.pyoracle import Oracle
        self.o = Oracle()
        self.K, self.V = K, V
        self.gamma = None

    def set_alpha(self, alpha):
        self.alpha = np.asarray(alpha, dtype=np.float64).copy()

    def set_lambda(self, lam):
        self.lam = np.asfortranarray(lam, dtype=np.float64).copy(order="F")

    def get_lambda(self):
        return self.lam.copy(order="F")

    def upload(self, csr):
        return csr

    def snapshot_lambda(self):
        self.lam_prime = self.lam.copy(order="F")

    def wordcounts(self, batch):
        wc = np.zeros(self.V)
        np.add.at(wc, batch.ids, batch.cnts.astype(np.float64))
        return torch.from_numpy(wc)

    def tr_init(self, wc, rho, eta, coef):
        add = rho * (eta + coef * wc.numpy())
        self.lam = np.asfortranarray((1. - rho) * self.lam_prime + add[None, :])

    def estep(self, batch, gamma0, max_iter, threshold):
        if gamma0 is not None:
            self.gamma = np.asfortranarray(gamma0)
        g, s, _ = self.o.estep(self.lam, self.alpha, batch.indptr, batch.ids, batch.cnts,
                               self.gamma, max_iter, threshold)
        self.gamma = g
        return torch.from_numpy(np.ascontiguousarray(s.ravel(order="F")))

    def blend(self, sstats, rho, eta, scale):
        s = sstats.numpy().reshape(self.K, self.V, order="F")
        self.last_sstats = s.copy(order="F")
        self.lam = self.o.mstep_blend(self.lam_prime, s, rho, eta, scale)

    # -- empirical Bayes / adaptive rate: this rank's sums, plain NumPy + SciPy ---------------
    def psi_gamma_diff(self, num_local):
        from scipy.special import digamma
        if num_local == 0:
            return np.zeros(self.K), False
        g = self.gamma
        return (digamma(g) - digamma(g.sum(axis=0))[None, :]).sum(axis=1), False

    def lambda_psi_stats(self):
        from scipy.special import digamma
        return float(digamma(self.lam).sum()), self.lam.sum(axis=1)

    def adaptive_stats(self, eta, scale, tau, internal):
        assert not internal
        upd = (eta + scale * self.last_sstats) - self.lam_prime
        if not hasattr(self, "ada"):
            self.ada = np.zeros_like(upd)
        self.ada = (1. - 1. / tau) * self.ada + 1. / tau * upd
        return float((upd * upd).sum()), float((self.ada * self.ada).sum())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


CASES = [dict(max_iter_tr=2, init_gamma=True, rho=-1.), dict(max_iter_tr=0, init_gamma=True, rho=.2),
         dict(max_iter_tr=3, init_gamma=False, rho=-1.),
         dict(max_iter_tr=2, init_gamma=True, rho=-1., presharded=True)]
K, V, D, B = 6, 120, 500, 23


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import trlda_amd
        from trlda_amd.distributed import ShardedOnlineLDA
        from trlda_amd.documents import CSRDocuments
        from trlda_amd.utils.synthetic import make_corpus
        for c, case in enumerate(CASES):
            case = dict(case)
            presharded = case.pop("presharded", False)
            # only rank 0's seed counts: the constructor hands its lambda AND its generator
            # state to the other ranks (the library seeds itself from the clock otherwise)
            trlda_amd.seed(500 + c if rank == 0 else 9000 + 10 * c + rank)
            model = ShardedOnlineLDA(V, K, D, alpha=.1, eta=.3, engine=OracleEngine(V, K))
            assert model.world == world and model.rank == rank
            assert model.replicas_agree()
            rhos = []
            for i in range(2):
                docs = CSRDocuments(*make_corpus(B, V, seed=40 + i, mean_unique=25))
                if presharded:                            # every rank brings its own documents
                    cuts = docs.shard_cuts(world)
                    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
                    rhos.append(model.update_parameters(docs.slice(lo, hi), max_iter_inference=20,
                                                        presharded=True, total_docs=B,
                                                        doc_range=(lo, hi), **case))
                else:
                    rhos.append(model.update_parameters(docs, max_iter_inference=20, **case))
            assert model.update_parameters([]) == 1.0 and model.update_count == 2
            assert model.replicas_agree()
            # lambda must be replicated: identical on every rank
            lam = torch.from_numpy(np.ascontiguousarray(model.lambdas))
            gathered = [torch.empty_like(lam) for _ in range(world)]
            dist.all_gather(gathered, lam)
            for g in gathered:
                assert torch.equal(g, gathered[0])
            if rank == 0:
                np.savez(os.path.join(outdir, "case%d.npz" % c), lam=model.lambdas,
                         rhos=np.array(rhos))
    finally:
        dist.destroy_process_group()


def _kwargs(arr):
    import ast
    return {str(k): ast.literal_eval(str(v)) for k, v in arr}


def _golden_worker(rank, world, port, outdir):
    """The empirical-Bayes / adaptive-rate trajectories of the COMPILED REFERENCE (golden f8:
    OnlineLDA, f9: BatchLDA) replayed by the sharded models over two gloo ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import trlda_amd
        from helpers import golden
        from trlda_amd.distributed import ShardedBatchLDA, ShardedOnlineLDA
        from trlda_amd.documents import CSRDocuments
        f = golden("f8_empirical_bayes")
        K8, V8, D8 = int(f["K"]), int(f["V"]), int(f["D"])
        res = {}
        for case in range(int(f["num_cases"])):
            kw = _kwargs(f["c%d_kwargs" % case])
            trlda_amd.seed(3000 + case if rank == 0 else 77 + rank)
            m = ShardedOnlineLDA(V8, K8, D8, alpha=np.linspace(.05, .4, K8), eta=.25,
                                 engine=OracleEngine(V8, K8))
            for i in range(4):
                docs = CSRDocuments(f["indptr%d" % i], f["ids%d" % i], f["cnts%d" % i])
                args = dict(max_iter_tr=2, max_iter_inference=20, kappa=.7, tau=10., rho=-1.)
                args.update(kw)
                if case % 2:                                  # odd cases: every rank its own shard
                    cuts = docs.shard_cuts(world)
                    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
                    r = m.update_parameters(docs.slice(lo, hi), presharded=True, total_docs=len(docs),
                                            doc_range=(lo, hi), **args)
                else:
                    r = m.update_parameters(docs, **args)
                assert m.replicas_agree()
                res["on%d_%d" % (case, i)] = np.concatenate(
                    [[r, m.eta], m.alpha.ravel(), m.lambdas.ravel(order="F")])
            assert m.update_count == 4
        f = golden("f9_batch_empirical_bayes")
        K9, V9 = int(f["K"]), int(f["V"])
        docs = CSRDocuments(f["indptr"], f["ids"], f["cnts"])
        for case in range(int(f["num_cases"])):
            kw = _kwargs(f["c%d_kwargs" % case])
            trlda_amd.seed(4000 + case if rank == 0 else 99 + rank)
            m = ShardedBatchLDA(V9, K9, alpha=np.linspace(.1, .6, K9), eta=.2,
                                engine=OracleEngine(V9, K9))
            args = dict(max_epochs=3, max_iter_inference=50)
            args.update(kw)
            assert m.update_parameters(docs, **args) == 1.
            assert m.replicas_agree()
            res["ba%d" % case] = np.concatenate([[m.eta], m.alpha.ravel(), m.lambdas.ravel(order="F")])
        # plain BatchLDA epochs (no empirical Bayes), every rank its own shard
        trlda_amd.seed(4100 if rank == 0 else 5)
        m = ShardedBatchLDA(V9, K9, alpha=.2, eta=.2, engine=OracleEngine(V9, K9))
        cuts = docs.shard_cuts(world)
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        m.update_parameters(docs.slice(lo, hi), max_epochs=2, max_iter_inference=30, presharded=True,
                            total_docs=len(docs), doc_range=(lo, hi))
        assert m.replicas_agree() and m.update_parameters([]) == 1.
        res["plain"] = m.lambdas.ravel(order="F")
        if rank == 0:
            np.savez(os.path.join(outdir, "golden_replay.npz"), **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_empirical_bayes_and_batch_lda_match_the_reference(tmp_path, oracle, hip_lib):
    """ShardedOnlineLDA with update_alpha / update_eta / adaptive and ShardedBatchLDA with its line
    searches, two gloo ranks: alpha, eta, lambda and rho after every call equal the compiled
    reference's single-process trajectories (golden f8 / f9; onlinelda.cpp:116-175,
    batchlda.cpp:43-205)."""
    from helpers import golden, relerr
    world = 2
    mp.spawn(_golden_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "golden_replay.npz"))
    f = golden("f8_empirical_bayes")
    K8 = int(f["K"])
    for case in range(int(f["num_cases"])):
        for i in range(4):
            g = got["on%d_%d" % (case, i)]
            assert abs(g[0] - f["c%d_rhos" % case][i]) <= 1e-9 * abs(g[0]), (case, i)
            assert abs(g[1] - float(f["c%d_eta%d" % (case, i + 1)])) < 1e-8 * g[1], (case, i)
            assert relerr(g[2:2 + K8], f["c%d_alpha%d" % (case, i + 1)].ravel()) < 1e-8, (case, i)
            assert relerr(g[2 + K8:], f["c%d_lambda%d" % (case, i + 1)].ravel(order="F")) < 1e-8, (case, i)
    f = golden("f9_batch_empirical_bayes")
    K9, V9 = int(f["K"]), int(f["V"])
    for case in range(int(f["num_cases"])):
        g = got["ba%d" % case]
        assert abs(g[0] - float(f["c%d_eta" % case])) < 1e-7 * g[0], case
        assert relerr(g[1:1 + K9], f["c%d_alpha" % case].ravel()) < 1e-7, case
        assert relerr(g[1 + K9:], f["c%d_lambda" % case].ravel(order="F")) < 1e-7, case
    # the plain epochs against the oracle's BatchLDA from the same stream
    oracle.seed(4100)
    lam = oracle.sample_gamma(K9, V9, 100) / 100.
    _, lam, _ = oracle.batch_update_parameters(lam, .2, .2, f["indptr"], f["ids"], f["cnts"],
                                               max_epochs=2, max_iter_inference=30)
    assert relerr(got["plain"], lam.ravel(order="F")) < 1e-11


@pytest.mark.timeout(300)
def test_two_rank_update_matches_single_process(tmp_path, oracle, hip_lib):
    from helpers import relerr, seeded_lambda
    from trlda_amd.utils.synthetic import make_corpus
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for c, case in enumerate(CASES):
        case = {k: v for k, v in case.items() if k != "presharded"}
        got = np.load(os.path.join(str(tmp_path), "case%d.npz" % c))
        # single-process trajectory from the same libc stream (ctor draws lambda, then gammas)
        oracle.seed(500 + c)
        lam = oracle.sample_gamma(K, V, 100) / 100.
        count = 0
        for i in range(2):
            ip, ii, cc = make_corpus(B, V, seed=40 + i, mean_unique=25)
            r, lam, count, _ = oracle.online_update_parameters(
                lam, .1, .3, D, ip, ii, cc, count, max_iter_tr=case["max_iter_tr"],
                max_iter_inference=20, rho=case["rho"], init_gamma=case["init_gamma"])
            assert abs(r - got["rhos"][i]) < 1e-15
        assert relerr(got["lam"], lam) < 1e-11


def test_single_rank_without_process_group(oracle, hip_lib):
    """world_size 1 (no init_process_group): same code path, no collective."""
    import trlda_amd
    from helpers import relerr
    from trlda_amd.distributed import ShardedOnlineLDA
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    assert not dist.is_initialized()
    trlda_amd.seed(77)
    model = ShardedOnlineLDA(V, K, D, engine=OracleEngine(V, K))
    ip, ii, cc = make_corpus(B, V, seed=1, mean_unique=25)
    r = model.update_parameters(CSRDocuments(ip, ii, cc), max_iter_tr=2)
    oracle.seed(77)
    lam = oracle.sample_gamma(K, V, 100) / 100.
    r2, lam, count, _ = oracle.online_update_parameters(lam, .1, .3, D, ip, ii, cc, 0,
                                                        max_iter_tr=2)
    assert r == r2 and model.update_count == 1
    assert relerr(model.lambdas, lam) < 1e-12


def test_exchange_choice_follows_the_bytes():
    """ShardedOnlineLDA.use_factors: the factor exchange where world * (docs_r K + nnz_r) doubles
    are fewer than the all-reduce's 2 K V (BASELINE.json configs 3 and 5), the all-reduce where
    documents outnumber words (config 4), and never without an engine that implements it."""
    from trlda_amd.distributed import ShardedOnlineLDA
    from trlda_amd.documents import CSRDocuments

    class Probe(ShardedOnlineLDA):
        def __init__(self, K, V, world, capable, exchange="auto"):
            self._K, self._V, self.world = K, V, world
            self._factors_ok, self.exchange = capable, exchange

    def even(B, n):
        return CSRDocuments(np.arange(B + 1, dtype=np.int32) * n, np.zeros(B * n, dtype=np.int32),
                            np.ones(B * n, dtype=np.int32))

    def choice(K, V, B, n, world, **kw):
        csr = even(B, n)
        return Probe(K, V, world, True, **kw).use_factors(csr, csr.shard_cuts(world))

    assert choice(100, 7000, 1600, 90, 8)            # config 3: 0.3 MB per rank against 5.6 MB
    assert choice(500, 100000, 4096, 100, 8)         # config 5: 2.5 MB per rank against 400 MB
    assert not choice(200, 50000, 100000, 100, 8)    # config 4: 240 MB gathered against 80 MB
    assert choice(200, 50000, 100000, 100, 8, exchange="factors")
    assert not choice(100, 7000, 1600, 90, 8, exchange="sstats")
    csr = even(16, 5)
    assert not Probe(10, 100, 2, False).use_factors(csr, csr.shard_cuts(2))
