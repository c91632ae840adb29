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
        self.lam = self.o.mstep_blend(self.lam_prime, s, rho, eta, scale)


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
