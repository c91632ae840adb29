"""Data parallelism with FACTOR exchange (csrc/dp_kernels.h, include/trlda_hip.h: trlda_model_*_dp):
the statistics of a mini-batch (reference src/lda.cpp:207-217) formed on every rank from an
all-gather of the documents' expElogtheta rows and per-entry weights instead of an all-reduce of
K x V numbers.

The GPU box has one device and RCCL refuses two ranks on it, so world > 1 runs as one PROCESS per
rank on that device (tests/dp_worker.py) with the all-gather hook as the transport; world = 1
goes through the same code with nothing to exchange.  Checked: every rank ends with bitwise the
same lambda, which equals the one-GPU update of the whole mini-batch (bitwise when the shards
select the document-kernel variant the whole batch selects), at the oracle's values.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import TIGHT_RTOL, relerr

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hip(hip_lib):
    from trlda_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a visible MI355X"
    return hip_lib


def corpus(B, V, seed, mean_unique=60, lengths=None):
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    return CSRDocuments(*make_corpus(B, V, seed=seed, mean_unique=mean_unique, lengths=lengths))


def random_lambda(K, V, seed):
    rng = np.random.RandomState(seed)
    return np.asfortranarray(rng.gamma(100., .01, (K, V)))


def run_ranks(tmp_path, cfg, world):
    """`world` worker processes on the one GPU; returns their result files"""
    path = str(tmp_path / "dp")
    cfg = dict(cfg, world=world, path=path)
    np.memmap(path + ".bar", dtype=np.int64, mode="w+", shape=(world,)).flush()
    np.memmap(path + ".dat", dtype=np.float64, mode="w+", shape=(world, cfg["max_count"])).flush()
    np.memmap(path + ".lam", dtype=np.float64, mode="w+", shape=(cfg["K"] * cfg["V"],)).flush()
    cfg_path = str(tmp_path / "cfg.json")
    json.dump(cfg, open(cfg_path, "w"))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), cfg_path, str(r)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "DP-RANK-OK" in so, (r, so[-1500:], se[-3000:])
    # (read now: a later run in the same directory rewrites the files)
    out = []
    for r in range(world):
        with np.load(path + ".rank%d.npz" % r) as f:
            out.append({k: f[k] for k in f.files})
    return out


class Single(object):
    """the one-GPU model through the C ABI"""

    def __init__(self, L, K, V, lam, alpha):
        from trlda_amd import _ffi
        self.L, self.K, self.V, self.ffi = L, K, V, _ffi
        self.h = _ffi.vp()
        _ffi.check(L.trlda_model_create(C.byref(self.h), 0, K, V))
        _ffi.check(L.trlda_model_set_lambda(self.h, lam))
        _ffi.check(L.trlda_model_set_alpha(self.h, np.full(K, alpha)))
        # (the statistics as a kernel of their own, as every *_dp call has them: an exchange sits
        # between its documents and its statistics -- csrc/estep_merged.h adds the row sums up in
        # another order, ~1e-12 in lambda after a few calls)
        _ffi.check(L.trlda_model_set_merged_launch(self.h, 0))
        self.count = C.c_int(0)

    def update(self, csr, D, eta, seed, max_iter_tr, max_iter_inference):
        from trlda_amd.documents import DeviceBatch
        b = DeviceBatch(csr, self.V, 0)
        self.L.trlda_seed(seed)
        rho = C.c_double(0.)
        self.ffi.check(self.L.trlda_model_online_update(
            self.h, b.handle, D, eta, max_iter_tr, max_iter_inference, .7, 100., -1., 1, 1, 1e-3,
            C.byref(self.count), C.byref(rho), None))
        self.ffi.check(self.L.trlda_model_synchronize(self.h))
        b.close()
        return rho.value

    def batch_update(self, csr, eta, seed, max_epochs, max_iter_inference):
        from trlda_amd.documents import DeviceBatch
        b = DeviceBatch(csr, self.V, 0)
        self.L.trlda_seed(seed)
        self.ffi.check(self.L.trlda_model_batch_update(self.h, b.handle, eta, max_epochs,
                                                       max_iter_inference, 1, 1e-3, None))
        b.close()

    def lambdas(self):
        lam = np.empty((self.K, self.V), order="F")
        self.ffi.check(self.L.trlda_model_get_lambda(self.h, lam))
        return lam

    def close(self):
        self.L.trlda_model_destroy(self.h)


def slot_bound(csrs, K, world):
    worst = 0
    for csr in csrs:
        cuts = csr.shard_cuts(world)
        docs = int(np.max(np.diff(cuts)))
        nnz = int(np.max(np.diff(csr.indptr[cuts])))
        worst = max(worst, (docs * K + nnz + K - 1) // K * K)
    return worst


@pytest.mark.parametrize("world", [2, 4])
def test_estep_factor_exchange_equals_one_gpu(hip, oracle, tmp_path, world):
    """One E-step: gamma, iteration counts and the K x V statistics of `world` ranks against the
    one-GPU E-step of the whole mini-batch and the oracle."""
    from trlda_amd import _ffi
    K, V, B = 100, 3000, 90
    csr = corpus(B, V, seed=811)
    lam = random_lambda(K, V, 17)
    g0 = np.asfortranarray(np.random.RandomState(3).gamma(100., .01, (K, B)))
    np.save(str(tmp_path / "g0.npy"), g0)
    cfg = dict(K=K, V=V, D=5000, alpha=.1, eta=.3, lambda_seed=17, max_count=slot_bound([csr], K, world),
               calls=[dict(kind="estep", B=B, corpus_seed=811, gamma0=str(tmp_path / "g0.npy"),
                           max_iter=20)])
    res = run_ranks(tmp_path, cfg, world)
    cuts = csr.shard_cuts(world)

    g_ref, s_ref, it_ref = oracle.estep(lam, np.full(K, .1), csr.indptr, csr.ids, csr.cnts, g0, 20,
                                        1e-3, nthreads=8)
    gamma = np.concatenate([r["gamma0"] for r in res], axis=1)
    iters = np.concatenate([r["iters0"] for r in res])
    assert gamma.shape == (K, B)
    assert np.array_equal(iters, it_ref)
    assert relerr(gamma, g_ref) < TIGHT_RTOL
    for r in res:
        assert np.array_equal(r["sstats0"], res[0]["sstats0"])      # replicas: bitwise
        assert int(r["exchanges"][0]) == 1
    assert relerr(res[0]["sstats0"], s_ref, floor=1e-12) < TIGHT_RTOL
    # mass balance (a17): sum of the statistics = sum of the counts
    assert abs(res[0]["sstats0"].sum() - csr.cnts.sum()) < 1e-8 * csr.cnts.sum()

    # the one-GPU run of the whole mini-batch (same document-kernel variant here: no document
    # of this corpus is longer than 128 words): bitwise
    assert int(np.max(np.diff(csr.indptr))) <= 128
    g_one = np.asfortranarray(g0.copy())
    s_one = np.empty((K, V), order="F")
    _ffi.check(hip.trlda_estep(K, V, B, csr.indptr, csr.ids, csr.cnts, lam, np.full(K, .1), g_one,
                               s_one, 20, 1e-3, None, 0))
    assert np.array_equal(g_one, gamma)
    assert np.array_equal(s_one, res[0]["sstats0"])
    assert int(cuts[-1]) == B


@pytest.mark.parametrize("world,max_iter_tr", [(2, 3), (3, 0)])
def test_online_update_dp_processes(hip, tmp_path, world, max_iter_tr):
    """OnlineLDA::updateParameters (onlinelda.cpp:53-111) twice over `world` ranks: every rank's
    lambda bitwise equal, and equal to the one-GPU model's; gamma0 comes from every rank's own
    libc stream, column range by column range."""
    K, V, D = 100, 4000, 20000
    specs = [dict(kind="update", B=70, corpus_seed=821, seed=5, max_iter_tr=max_iter_tr,
                  max_iter_inference=20),
             dict(kind="update", B=51, corpus_seed=822, seed=6, max_iter_tr=max_iter_tr,
                  max_iter_inference=20)]
    csrs = [corpus(s["B"], V, seed=s["corpus_seed"]) for s in specs]
    cfg = dict(K=K, V=V, D=D, alpha=.1, eta=.3, lambda_seed=23,
               max_count=slot_bound(csrs, K, world), calls=specs)
    res = run_ranks(tmp_path, cfg, world)

    one = Single(hip, K, V, random_lambda(K, V, 23), .1)
    rhos = [one.update(c, D, .3, s["seed"], max_iter_tr, 20) for c, s in zip(csrs, specs)]
    lam_one = one.lambdas()
    one.close()
    for r in res:
        assert np.array_equal(r["lambda"], res[0]["lambda"])
        assert [float(r["rho0"][0]), float(r["rho1"][0])] == rhos
        assert int(r["update_count"][0]) == 2
        # one exchange per E-step, and nothing else crosses ranks (no word-count all-reduce)
        assert int(r["exchanges"][0]) == 2 * max(max_iter_tr, 1)
    assert np.array_equal(res[0]["lambda"], lam_one) or relerr(res[0]["lambda"], lam_one) < 1e-12
    assert relerr(res[0]["lambda"], lam_one) < 1e-12


@pytest.mark.parametrize("world", [2, 3, 4])
def test_word_sharded_m_step_processes(hip, tmp_path, world):
    """The word-sharded M-step (include/trlda_hip.h, trlda_model_set_allgatherv): after the factor
    exchange rank r forms statistics + M-step for ITS range of the vocabulary only and the ranks
    exchange the lambda columns they wrote, in place.  Two updateParameters calls with a
    trust-region loop, one without, and two BatchLDA epochs, over 2 / 3 / 4 ranks as processes:
    every rank's lambda bitwise equal; equal to the run in which every rank forms the whole
    mini-batch's statistics and to the one-GPU model (the columns bitwise after the first M-step;
    later E-steps see row sums added up in another order: 1e-12); one factor exchange per E-step,
    one lambda exchange per M-step, nothing else."""
    K, V, D = 100, 4000, 20000
    specs = [dict(kind="update", B=90, corpus_seed=871, seed=5, max_iter_tr=3, max_iter_inference=20),
             dict(kind="update", B=64, corpus_seed=872, seed=6, max_iter_tr=0, max_iter_inference=20),
             dict(kind="update", B=75, corpus_seed=873, seed=7, max_iter_tr=2, max_iter_inference=20),
             dict(kind="batch", B=80, corpus_seed=874, seed=8, max_epochs=2, max_iter_inference=20)]
    csrs = [corpus(s["B"], V, seed=s["corpus_seed"]) for s in specs]
    cfg = dict(K=K, V=V, D=D, alpha=.1, eta=.3, lambda_seed=53,
               max_count=slot_bound(csrs, K, world), calls=specs)
    res = run_ranks(tmp_path, cfg, world)
    whole = run_ranks(tmp_path, dict(cfg, word_sharded=False), world)
    e_steps = 3 + 1 + 2 + 2
    for r in res:
        assert int(r["word_sharded"][0]) == 1
        assert np.array_equal(r["lambda"], res[0]["lambda"])          # replicas: bitwise
        assert int(r["exchanges"][0]) == e_steps and int(r["lambda_exchanges"][0]) == e_steps
        # what a rank receives per M-step: the table minus its own range
        assert 0 < int(r["lambda_exchanges"][1]) < e_steps * K * V * 8
    for r in whole:
        assert int(r["word_sharded"][0]) == 0 and int(r["lambda_exchanges"][0]) == 0
        assert np.array_equal(r["lambda"], whole[0]["lambda"])
    assert relerr(res[0]["lambda"], whole[0]["lambda"]) < 1e-11
    one = Single(hip, K, V, random_lambda(K, V, 53), .1)
    rhos = [one.update(c, D, .3, s["seed"], s["max_iter_tr"], 20) for c, s in zip(csrs[:3], specs[:3])]
    assert [float(res[0]["rho%d" % i][0]) for i in range(3)] == rhos
    one.batch_update(csrs[3], .3, 8, 2, 20)
    assert relerr(res[0]["lambda"], one.lambdas()) < 1e-11
    one.close()
    # one call, one M-step (no trust-region loop): nothing downstream of the M-step -- bitwise the
    # one-GPU lambda, whoever computed a column
    first = run_ranks(tmp_path, dict(cfg, calls=specs[1:2]), world)
    one = Single(hip, K, V, random_lambda(K, V, 53), .1)
    one.update(csrs[1], D, .3, 6, 0, 20)
    assert np.array_equal(first[0]["lambda"], one.lambdas())
    assert int(first[0]["word_sharded"][0]) == 1
    one.close()


def test_config3_literally_eight_ranks(hip, tmp_path):
    """BASELINE.json config 3 as it is worded: OnlineLDA K = 100, V = 7000, a mini-batch of 1600
    documents sharded 200 per rank over EIGHT ranks (here: eight processes on the one GPU, the
    all-gather hook as transport), the default exchange plan -- factors all-gathered, statistics +
    M-step sharded by vocabulary range, lambda columns exchanged in place -- with the trust-region
    loop (max_iter_tr = 3) and without (0): eight bitwise-equal replicas; the call without a loop
    bitwise the one-GPU lambda of the whole 1600-document mini-batch, the trajectory to 1e-11;
    one factor exchange per E-step, one lambda exchange per M-step.  (the
    world-8 shape had never run, even as processes.)"""
    K, V, D, world = 100, 7000, 1000000, 8
    specs = [dict(kind="update", B=1600, corpus_seed=941, seed=15, max_iter_tr=0, max_iter_inference=20,
                  mean_unique=100),
             dict(kind="update", B=1600, corpus_seed=942, seed=16, max_iter_tr=3, max_iter_inference=20,
                  mean_unique=100)]
    csrs = [corpus(1600, V, seed=s["corpus_seed"], mean_unique=s["mean_unique"]) for s in specs]
    for c in csrs:                                   # 200 documents per rank, to a few by nnz balance
        assert np.all(np.abs(np.diff(c.shard_cuts(world)) - 200) <= 12)
    cfg = dict(K=K, V=V, D=D, alpha=.1, eta=.3, lambda_seed=61,
               max_count=slot_bound(csrs, K, world), calls=specs)
    # the call without a trust-region loop on its own: one M-step, nothing downstream of it
    first = run_ranks(tmp_path, dict(cfg, calls=specs[:1]), world)
    one = Single(hip, K, V, random_lambda(K, V, 61), .1)
    rho0 = one.update(csrs[0], D, .3, 15, 0, 20)
    for r in first:
        assert int(r["word_sharded"][0]) == 1
        assert np.array_equal(r["lambda"], first[0]["lambda"])
        assert int(r["exchanges"][0]) == 1 and int(r["lambda_exchanges"][0]) == 1
    assert float(first[0]["rho0"][0]) == rho0
    # bitwise: every column computed once, by its owner, in document order
    assert np.array_equal(first[0]["lambda"], one.lambdas())
    # both calls: 1 + 3 E-steps
    res = run_ranks(tmp_path, cfg, world)
    rho1 = one.update(csrs[1], D, .3, 16, 3, 20)
    for r in res:
        assert int(r["word_sharded"][0]) == 1
        assert np.array_equal(r["lambda"], res[0]["lambda"])          # replicas: bitwise
        assert int(r["exchanges"][0]) == 4 and int(r["lambda_exchanges"][0]) == 4
        # per M-step a rank receives the table minus its own range (ranges are balanced by entries,
        # not by words: the owner of the rare words holds many columns)
        assert 0 < int(r["lambda_exchanges"][1]) < 4 * K * V * 8
    # ... over all ranks: 7/8 of the table per rank and M-step
    assert sum(int(r["lambda_exchanges"][1]) for r in res) == 4 * (world - 1) * K * V * 8
    for r in res:
        pass
    assert [float(res[0]["rho%d" % i][0]) for i in range(2)] == [rho0, rho1]
    assert relerr(res[0]["lambda"], one.lambdas()) < 1e-11
    one.close()


def test_online_update_dp_large_table_and_plain_sequence(hip, tmp_path):
    """K = 500 (the one-orientation document kernel, streaming row sums, the pair-gather
    statistics kernel) over two ranks; and the plain launch sequence (fused update off) through
    the same exchange."""
    K, V, D = 500, 12000, 50000
    specs = [dict(kind="update", B=48, corpus_seed=831, seed=9, max_iter_tr=2, max_iter_inference=20)]
    csrs = [corpus(48, V, seed=831)]
    lam_one = None
    for plain in (0, 1):
        cfg = dict(K=K, V=V, D=D, alpha=.1, eta=.3, lambda_seed=29, plain=plain,
                   max_count=slot_bound(csrs, K, 2), calls=specs)
        res = run_ranks(tmp_path, cfg, 2)
        assert np.array_equal(res[0]["lambda"], res[1]["lambda"])
        if lam_one is None:
            one = Single(hip, K, V, random_lambda(K, V, 29), .1)
            one.update(csrs[0], D, .3, 9, 2, 20)
            lam_one = one.lambdas()
            one.close()
        assert relerr(res[0]["lambda"], lam_one) < 1e-11, (plain, relerr(res[0]["lambda"], lam_one))


@pytest.mark.parametrize("world", [2, 3])
def test_direct_slot_exchange_processes(hip, oracle, tmp_path, world):
    """The direct exchange (trlda_model_dp_direct_alloc / _connect): every rank writes its slot
    into its peers' gather buffers through hipIpc-mapped pointers and signals them with a step
    counter -- no all-gather hook, no collective.  Ranks as processes on the one GPU (IPC works
    on the same device).  One E-step and two updateParameters calls (5 and 1 exchanges): every
    rank's statistics and lambda bitwise equal, equal to the one-GPU results and the oracle's."""
    K, V, D, B = 100, 4000, 20000, 75
    # the region must be fine-grained device memory that hipIpc exports (no fallback to ordinary
    # memory any more): a box whose runtime refuses that has no direct exchange to test
    from trlda_amd import _ffi
    probe, handle = _ffi.vp(), C.create_string_buffer(64)
    _ffi.check(hip.trlda_model_create(C.byref(probe), 0, 4, 8))
    rc = hip.trlda_model_dp_direct_alloc(probe, 64, world, handle)
    why = hip.trlda_last_error()
    hip.trlda_model_destroy(probe)
    if rc != 0:
        pytest.skip("direct exchange unavailable here: %s" % (why.decode() if isinstance(why, bytes) else why))
    csr = corpus(B, V, seed=881)
    lam = random_lambda(K, V, 43)
    g0 = np.asfortranarray(np.random.RandomState(6).gamma(100., .01, (K, B)))
    np.save(str(tmp_path / "g0.npy"), g0)
    specs = [dict(kind="estep", B=B, corpus_seed=881, gamma0=str(tmp_path / "g0.npy"), max_iter=20),
             dict(kind="update", B=B, corpus_seed=881, seed=13, max_iter_tr=4, max_iter_inference=20),
             dict(kind="update", B=60, corpus_seed=882, seed=14, max_iter_tr=0, max_iter_inference=20)]
    csrs = [csr, csr, corpus(60, V, seed=882)]
    cfg = dict(K=K, V=V, D=D, alpha=.1, eta=.3, lambda_seed=43, direct=True,
               max_count=slot_bound(csrs, K, world) + 64, calls=specs)
    res = run_ranks(tmp_path, cfg, world)
    g_ref, s_ref, it_ref = oracle.estep(lam, np.full(K, .1), csr.indptr, csr.ids, csr.cnts, g0, 20,
                                        1e-3, nthreads=8)
    assert np.array_equal(np.concatenate([r["iters0"] for r in res]), it_ref)
    assert relerr(np.concatenate([r["gamma0"] for r in res], axis=1), g_ref) < TIGHT_RTOL
    for r in res:
        assert np.array_equal(r["sstats0"], res[0]["sstats0"])
        assert np.array_equal(r["lambda"], res[0]["lambda"])
        assert int(r["exchanges"][0]) == 0                   # the hook transport never ran
    assert relerr(res[0]["sstats0"], s_ref, floor=1e-12) < TIGHT_RTOL
    one = Single(hip, K, V, lam, .1)
    rhos = [one.update(csrs[1], D, .3, 13, 4, 20), one.update(csrs[2], D, .3, 14, 0, 20)]
    assert [float(res[0]["rho1"][0]), float(res[0]["rho2"][0])] == rhos
    assert relerr(res[0]["lambda"], one.lambdas()) < 1e-12
    one.close()


def test_long_document_in_one_shard_and_an_empty_shard(hip, oracle, tmp_path):
    """The ranks must agree on what the gathered factors mean whatever their own shards hold
    (ADVICE r2): one shard with a 300-word document (beyond the register kernel), one EMPTY shard,
    one ordinary shard -- every rank's statistics and lambda bitwise equal, the mass balance
    holds, and the results equal the one-GPU call's and the oracle's."""
    from trlda_amd import _ffi
    K, V, B, D, world = 100, 3000, 60, 8000, 3
    lengths = [40 + (7 * i) % 50 for i in range(B)]
    lengths[3] = 300                                        # in shard 0
    cuts = [0, 25, 25, B]                                   # rank 1 holds nothing
    csr = corpus(B, V, seed=861, lengths=lengths)
    lam = random_lambda(K, V, 37)
    g0 = np.asfortranarray(np.random.RandomState(5).gamma(100., .01, (K, B)))
    np.save(str(tmp_path / "g0.npy"), g0)
    specs = [dict(kind="estep", B=B, corpus_seed=861, lengths=lengths, cuts=cuts,
                  gamma0=str(tmp_path / "g0.npy"), max_iter=20),
             dict(kind="update", B=B, corpus_seed=861, lengths=lengths, cuts=cuts, seed=11,
                  max_iter_tr=2, max_iter_inference=20)]
    cfg = dict(K=K, V=V, D=D, alpha=.1, eta=.3, lambda_seed=37,
               max_count=(35 * K + int(csr.indptr[-1]) + K), calls=specs)
    res = run_ranks(tmp_path, cfg, world)
    g_ref, s_ref, it_ref = oracle.estep(lam, np.full(K, .1), csr.indptr, csr.ids, csr.cnts, g0, 20,
                                        1e-3, nthreads=8)
    assert res[1]["gamma0"].shape[1] == 0
    gamma = np.concatenate([r["gamma0"] for r in res], axis=1)
    assert np.array_equal(np.concatenate([r["iters0"] for r in res]), it_ref)
    assert relerr(gamma, g_ref) < TIGHT_RTOL
    for r in res:
        assert np.array_equal(r["sstats0"], res[0]["sstats0"])
        assert np.array_equal(r["lambda"], res[0]["lambda"])
    assert relerr(res[0]["sstats0"], s_ref, floor=1e-12) < TIGHT_RTOL
    assert abs(res[0]["sstats0"].sum() - csr.cnts.sum()) < 1e-8 * csr.cnts.sum()
    one = Single(hip, K, V, lam, .1)
    rho = one.update(csr, D, .3, 11, 2, 20)
    assert float(res[0]["rho1"][0]) == rho
    assert relerr(res[0]["lambda"], one.lambdas()) < 1e-12
    one.close()


@pytest.mark.parametrize("K,V", [(100, 4000), (200, 9000)])
def test_batch_update_dp_processes(hip, tmp_path, K, V):
    """BatchLDA::updateParameters (batchlda.cpp:43-61), two epochs over two ranks with the factor
    exchange: lambda bitwise equal across ranks and equal to the one-GPU model's."""
    B = 80
    specs = [dict(kind="batch", B=B, corpus_seed=871, seed=21, max_epochs=2, max_iter_inference=30)]
    csrs = [corpus(B, V, seed=871)]
    cfg = dict(K=K, V=V, D=0, alpha=.1, eta=.3, lambda_seed=41, max_count=slot_bound(csrs, K, 2),
               calls=specs)
    res = run_ranks(tmp_path, cfg, 2)
    one = Single(hip, K, V, random_lambda(K, V, 41), .1)
    one.batch_update(csrs[0], .3, 21, 2, 30)
    lam_one = one.lambdas()
    one.close()
    assert np.array_equal(res[0]["lambda"], res[1]["lambda"])
    assert int(res[0]["exchanges"][0]) == 2
    assert relerr(res[0]["lambda"], lam_one) < 1e-12
    # lambda = eta + sstats: the mass balance of the last epoch
    assert abs(lam_one.sum() - (.3 * K * V + csrs[0].cnts.sum())) < 1e-9 * lam_one.sum()


def test_world_one_is_the_single_gpu_call(hip):
    """trlda_model_online_update_dp at world = 1 == trlda_model_online_update, bit for bit."""
    from trlda_amd import _ffi
    from trlda_amd.documents import DeviceBatch
    L = hip
    K, V, D, B = 100, 5000, 30000, 120
    csr = corpus(B, V, seed=841, mean_unique=90)
    lam0 = random_lambda(K, V, 31)
    a = Single(L, K, V, lam0, .1)
    b = Single(L, K, V, lam0, .1)
    # (the one-GPU call with the statistics as a kernel of their own, as every *_dp call has them --
    # an exchange sits between its documents and its statistics; as workgroups of the document
    # launch, csrc/estep_merged.h, the row sums are added up in another order: ~1e-12 in lambda,
    # tests/test_gpu_merged.py)
    assert L.trlda_model_set_merged_launch(a.h, 0) == 0
    cuts = np.array([0, B], dtype=np.int32)
    for call, tr in enumerate((4, 0)):
        a.update(csr, D, .3, 40 + call, tr, 20)
        batch = DeviceBatch(csr, V, 0)
        L.trlda_seed(40 + call)
        rho = C.c_double(0.)
        _ffi.check(L.trlda_model_online_update_dp(
            b.h, batch.handle, batch.handle, None, 0, 1, cuts.ctypes.data_as(C.POINTER(C.c_int32)), D,
            .3, tr, 20, .7, 100., -1., 1, 1e-3, C.byref(b.count), C.byref(rho)))
        _ffi.check(L.trlda_model_synchronize(b.h))
        batch.close()
        assert np.array_equal(a.lambdas(), b.lambdas()), call
    a.close()
    b.close()


def test_dp_argument_checks(hip):
    from trlda_amd import _ffi
    from trlda_amd.documents import DeviceBatch
    L = hip
    K, V, B = 20, 300, 10
    csr = corpus(B, V, seed=851, mean_unique=20)
    m = Single(L, K, V, random_lambda(K, V, 1), .1)
    batch = DeviceBatch(csr, V, 0)
    shard = DeviceBatch(csr.slice(0, 4), V, 0)
    rho = C.c_double(0.)

    def call(cuts, rank, world, sh):
        cuts = np.asarray(cuts, dtype=np.int32)
        return L.trlda_model_online_update_dp(
            m.h, batch.handle, sh.handle, None, rank, world, cuts.ctypes.data_as(C.POINTER(C.c_int32)),
            100, .3, 0, 5, .7, 100., -1., 1, 1e-3, C.byref(m.count), C.byref(rho))

    assert call([0, 4, 9], 0, 2, shard) != 0           # cuts do not end at the batch size
    assert call([0, 1000, 10], 0, 2, shard) != 0       # a cut point beyond the batch
    assert call([0, 5, 10], 0, 2, shard) != 0          # shard is not documents [0, 5)
    assert call([0, 4, 10], 2, 2, shard) != 0          # rank outside the world
    assert call([0, 4, 10], 0, 2, shard) != 0          # two ranks, no communicator, no hook
    assert b"communicator" in L.trlda_last_error()
    assert m.count.value == 0
    batch.close()
    shard.close()
    m.close()
