"""One rank of a REAL data-parallel run: one process per GPU, torch.distributed backend "nccl"
(RCCL over xGMI), the sharded models of trlda_amd/distributed.py on top.  Test infrastructure for
tests/test_gpu_rccl.py, which starts `world` of these (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
the environment, as torch.distributed.run would).

usage: rccl_worker.py <config.json>
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    cfg = json.load(open(sys.argv[1]))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import trlda_amd
    from trlda_amd.distributed import ShardedBatchLDA, ShardedOnlineLDA
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=device)
    out = {}
    try:
        K, V, D = cfg["K"], cfg["V"], cfg["D"]
        for r, run in enumerate(cfg["runs"]):
            # only rank 0's seed counts (the constructor broadcasts its lambda and its stream)
            trlda_amd.seed(run["seed"] if rank == 0 else 100000 + 17 * rank + r)
            own = run.get("own_communicator", True)
            if run["model"] == "online":
                m = ShardedOnlineLDA(V, K, D, alpha=run.get("alpha", .1), eta=.3, device=local,
                                     exchange=run["exchange"], own_communicator=own,
                                     direct_exchange=run.get("direct", False))
            else:
                m = ShardedBatchLDA(V, K, alpha=run.get("alpha", .1), eta=.3, device=local,
                                    exchange=run["exchange"], own_communicator=own,
                                    direct_exchange=run.get("direct", False))
            assert m.world == world and m.rank == rank
            rhos, paths = [], []
            for call in run["calls"]:
                csr = CSRDocuments(*make_corpus(call["B"], V, seed=call["corpus_seed"],
                                                mean_unique=call.get("mean_unique", 60),
                                                lengths=call.get("lengths")))
                kw = dict(call.get("kwargs", {}))
                if call.get("presharded"):
                    cuts = csr.shard_cuts(world)
                    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
                    rhos.append(m.update_parameters(csr.slice(lo, hi), presharded=True,
                                                    total_docs=len(csr), doc_range=(lo, hi), **kw))
                else:
                    rhos.append(m.update_parameters(csr, **kw))
                paths.append(m.last_path)
            agree = m.replicas_agree()
            # bitwise: every rank's lambda against rank 0's
            lam = torch.from_numpy(np.ascontiguousarray(m.lambdas)).to(device)
            ref = lam.clone()
            dist.broadcast(ref, src=0)
            same = torch.tensor([int(torch.equal(lam, ref))], device=device)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if rank == 0:
                out["run%d_lambda" % r] = m.lambdas
                out["run%d_alpha" % r] = m.alpha.ravel()
                out["run%d_eta" % r] = np.array([m.eta])
                out["run%d_rhos" % r] = np.array(rhos)
                out["run%d_flags" % r] = np.array([int(agree), int(same.item()), int(m._own_comm),
                                                   int(getattr(m, "update_count", 0))])
                out["run%d_paths" % r] = np.array(paths, dtype=str)
                from trlda_amd import rccl
                out["run%d_rccl_ranks" % r] = np.array([rccl.comm_count(m.engine.comm)
                                                        if m.engine.comm else 0])
            m.close()
        if rank == 0:
            np.savez(cfg["out"], **out)
    finally:
        dist.destroy_process_group()
    print("RCCL-RANK-OK", rank)


if __name__ == "__main__":
    main()
