"""The data-parallel models over REAL RCCL: one process per GPU, backend "nccl", the sharded
models of trlda_amd/distributed.py with a ncclComm_t of the process's own underneath the C ABI's
whole-call entry points (include/trlda_hip.h: trlda_model_online_update_multi / _dp,
trlda_model_batch_update_multi / _dp, trlda_model_eb_gamma_stats_multi).

`world = min(device_count, 8)` when the box has at least two GPUs: factor exchange against the
all-reduce against the one-GPU result, bitwise-equal replicas, for OnlineLDA (with the
empirical-Bayes steps and the adaptive rate) and BatchLDA.  On a one-GPU box those tests SKIP and
the same worker runs at world 1 -- a real communicator of one rank, the same C entry points --
so that what the first multi-GPU node executes has run before.
Reference reduction points: src/lda.cpp:211-217, src/onlinelda.cpp:79-82, :128,
src/batchlda.cpp:43-61.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

K, V, D = 100, 5000, 40000
ONLINE_CALLS = [dict(B=96, corpus_seed=901, kwargs=dict(max_iter_tr=3, max_iter_inference=20)),
                dict(B=77, corpus_seed=902, kwargs=dict(max_iter_tr=0, max_iter_inference=20)),
                dict(B=64, corpus_seed=903, presharded=True,
                     kwargs=dict(max_iter_tr=2, max_iter_inference=20, update_alpha=True,
                                 update_eta=True, adaptive=True, tau=10.))]
# a document beyond the register kernel in one shard only: lengths of call 0
LONG = [50 + (11 * i) % 40 for i in range(96)]
LONG[5] = 260
BATCH_CALLS = [dict(B=120, corpus_seed=911, kwargs=dict(max_epochs=2, max_iter_inference=30)),
               dict(B=120, corpus_seed=911, presharded=True,
                    kwargs=dict(max_epochs=2, max_iter_inference=30, update_alpha=True,
                                update_eta=True))]


def runs():
    long_calls = [dict(ONLINE_CALLS[0], lengths=LONG)] + ONLINE_CALLS[1:]
    return [dict(model="online", exchange="factors", seed=31, calls=long_calls[:2]),
            dict(model="online", exchange="sstats", seed=31, calls=long_calls[:2]),
            dict(model="online", exchange="sstats", seed=31, calls=long_calls[:2],
                 own_communicator=False),
            dict(model="online", exchange="auto", seed=32, alpha=.2, calls=ONLINE_CALLS),
            dict(model="online", exchange="sstats", seed=32, alpha=.2, calls=ONLINE_CALLS),
            dict(model="batch", exchange="factors", seed=33, calls=BATCH_CALLS[:1]),
            dict(model="batch", exchange="sstats", seed=33, calls=BATCH_CALLS[:1]),
            dict(model="batch", exchange="sstats", seed=34, calls=BATCH_CALLS),
            # the direct slot exchange (hipIpc-mapped peer buffers, no collective per step)
            dict(model="online", exchange="factors", seed=31, calls=long_calls[:2], direct=True)]


@pytest.fixture(scope="module")
def hip(hip_lib):
    from trlda_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a visible MI355X"
    return hip_lib


def launch(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / ("rccl_world%d.npz" % world))
    cfg_path = str(tmp_path / ("cfg%d.json" % world))
    json.dump(dict(K=K, V=V, D=D, runs=runs(), out=out), open(cfg_path, "w"))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_worker.py"), cfg_path],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    try:
        outs = [p.communicate(timeout=1200) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "RCCL-RANK-OK" in so, (r, so[-1500:], se[-3000:])
    return np.load(out)


def one_gpu(hip):
    """The same runs on the single-GPU models (the yardstick every world size must meet)."""
    import trlda_amd
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.models import BatchLDA, OnlineLDA
    from trlda_amd.utils.synthetic import make_corpus
    res = []
    for run in runs():
        trlda_amd.seed(run["seed"])
        if run["model"] == "online":
            m = OnlineLDA(V, K, D, alpha=run.get("alpha", .1), eta=.3, device=0)
        else:
            m = BatchLDA(V, K, alpha=run.get("alpha", .1), eta=.3, device=0)
        rhos = []
        for call in run["calls"]:
            csr = CSRDocuments(*make_corpus(call["B"], V, seed=call["corpus_seed"],
                                            mean_unique=call.get("mean_unique", 60),
                                            lengths=call.get("lengths")))
            rhos.append(m.update_parameters(csr, **call.get("kwargs", {})))
        res.append(dict(lam=np.array(m.lambdas), alpha=m.alpha.ravel(), eta=m.eta, rhos=rhos))
        m.close()
    return res


def check(got, want, world):
    for r, (run, w) in enumerate(zip(runs(), want)):
        agree, same, own, count = [int(x) for x in got["run%d_flags" % r]]
        tag = (r, run["model"], run["exchange"], world)
        assert agree == 1 and same == 1, tag                  # replicas: bitwise equal lambda
        assert own == int(run.get("own_communicator", True)), tag
        if own:
            assert int(got["run%d_rccl_ranks" % r][0]) == world, tag
        paths = [str(p) for p in got["run%d_paths" % r]]
        if run["exchange"] == "factors":
            # (a presharded call cannot take the factor path: a rank only sees its shard; at one
            # rank there is nothing to exchange, directly or otherwise)
            want_path = "factors-direct" if run.get("direct") and world > 1 else "factors"
            assert all(p == want_path for p, c in zip(paths, run["calls"]) if not c.get("presharded")), (tag, paths)
        if run["exchange"] == "sstats":
            assert all(p == ("allreduce" if own else "allreduce-composed") for p in paths), (tag, paths)
        if run["model"] == "online":
            assert count == len(run["calls"]), tag
            assert np.allclose(got["run%d_rhos" % r], w["rhos"], rtol=1e-9, atol=0), tag
        # against the one-GPU model: the all-reduce's order of additions depends on the world
        # size, the factor path's does not (it is the one-GPU kernel on the whole mini-batch)
        assert relerr(got["run%d_lambda" % r], w["lam"]) < 1e-9, (tag, relerr(got["run%d_lambda" % r], w["lam"]))
        assert relerr(got["run%d_alpha" % r], w["alpha"]) < 1e-8, tag
        assert abs(float(got["run%d_eta" % r][0]) - w["eta"]) < 1e-8 * w["eta"], tag
    # the two exchanges against each other on the same run
    assert relerr(got["run0_lambda"], got["run1_lambda"]) < 1e-11
    assert relerr(got["run1_lambda"], got["run2_lambda"]) < 1e-11
    assert relerr(got["run3_lambda"], got["run4_lambda"]) < 1e-9
    assert relerr(got["run5_lambda"], got["run6_lambda"]) < 1e-11
    # direct == all-gather: bitwise at one rank; beyond that the all-gather run shards the M-step by
    # words (round 4: every rank its range of the vocabulary, lambda columns exchanged in place, the
    # next row sums from the exchanged table) while the direct run still forms the whole mini-batch's
    # statistics on every rank -- other row sums downstream, 1e-12
    if world == 1:
        assert np.array_equal(got["run8_lambda"], got["run0_lambda"])
    else:
        assert relerr(got["run8_lambda"], got["run0_lambda"]) < 1e-11


def test_rccl_world_one_exercises_the_same_entry_points(hip, tmp_path):
    """A real ncclComm_t of ONE rank under every whole-call entry point: what a multi-GPU node
    executes, minus the wire."""
    check(launch(tmp_path, 1), one_gpu(hip), 1)


def test_rccl_all_gpus_of_the_node(hip, tmp_path):
    """world = min(device_count, 8) ranks over RCCL / xGMI; auto-enabled on a multi-GPU box."""
    from trlda_amd import _ffi
    n = min(_ffi.device_count(), 8)
    if n < 2:
        pytest.skip("one GPU visible: the multi-rank RCCL run needs at least two")
    want = one_gpu(hip)
    check(launch(tmp_path, n), want, n)
    if n > 2:
        check(launch(tmp_path, 2), want, 2)
