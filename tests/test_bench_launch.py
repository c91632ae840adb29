"""`python bench.py --gpus N` with no launcher around it must start its N ranks itself -- as child
processes, before anything in the parent touches the GPU runtime -- relay exactly one JSON line,
and fail cleanly (non-zero, no JSON, no survivors) when a rank fails or hangs.  CPU only: the
ranks run bench.py's --dry-run-launch leg (no torch, no GPU)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(n, extra_env=None, timeout=60, args=()):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run-launch"] + list(args),
                          env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_spawns_every_rank_and_relays_one_line(tmp_path):
    r = run(4, {"TRLDA_BENCH_DRY_DIR": str(tmp_path)})
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["torch_in_rank"] is False
    seen = sorted(n for n in os.listdir(tmp_path) if n.startswith("rank"))
    assert seen == ["rank0", "rank1", "rank2", "rank3"]
    for i, name in enumerate(seen):
        rank, world, addr, parent_had_torch = open(tmp_path / name).read().split()
        # the rendezvous address the ranks get, and: the parent had not imported torch (hence
        # had not initialised any GPU runtime) when it started its ranks
        assert (int(rank), int(world), addr, parent_had_torch) == (i, 4, "127.0.0.1", "0")


def test_a_failing_rank_ends_the_others_without_a_result(tmp_path):
    t = time.time()
    r = run(3, {"TRLDA_BENCH_DRY_FAIL_RANK": "1", "TRLDA_BENCH_DRY_HANG_RANK": "2",
                "TRLDA_BENCH_DRY_DIR": str(tmp_path)})
    assert r.returncode != 0
    assert r.stdout.strip() == ""                     # no JSON line
    assert "rank 1 exited with status 7" in r.stderr
    assert time.time() - t < 30                       # the hanging rank was ended, not waited for


def test_a_hang_is_ended_by_the_deadline(tmp_path):
    t = time.time()
    r = run(2, {"TRLDA_BENCH_DRY_HANG_RANK": "1"}, args=["--launch-timeout", "2"])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "--launch-timeout" in r.stderr
    assert time.time() - t < 30


def test_under_a_launcher_it_does_not_spawn(tmp_path):
    # torch.distributed.run's environment: this process IS rank 1 of 2
    r = run(2, {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": "29999", "TRLDA_BENCH_DRY_DIR": str(tmp_path)})
    assert r.returncode == 0 and r.stdout.strip() == ""
    assert sorted(os.listdir(tmp_path)) == ["plan1.json", "rank1"]


def test_eight_ranks_build_the_same_exchange_plan(tmp_path):
    """`bench.py --gpus 8` (BASELINE.json config 3: K = 100, V = 7000, 200 documents per GPU): every
    rank, from its arguments alone, arrives at the same plan for what crosses ranks -- `auto` = the
    factor all-gather (1.4 MB per rank and step against 11.2 MB for the all-reduce) with the
    statistics + M-step sharded by vocabulary range; the direct (hipIpc) exchange only on request;
    the same at config 5's shape (K = 500, V = 100 000, 512 documents per GPU); the all-reduce
    where it moves fewer bytes (a tiny vocabulary under large batches)."""
    r = run(8, {"TRLDA_BENCH_DRY_DIR": str(tmp_path)})
    assert r.returncode == 0, r.stderr
    plans = [json.load(open(tmp_path / ("plan%d.json" % i))) for i in range(8)]
    assert all(p == plans[0] for p in plans)
    p = plans[0]
    assert p["exchange"] == "factors" and p["word_sharded_m_step"] is True and p["direct_requested"] is False
    assert p["batch_per_gpu"] == 200 and p["factors_bytes_per_rank"] < p["allreduce_bytes_per_rank"]
    assert json.loads(r.stdout)["plan"] == p
    # ... and at the same list of plans to time one after the other in that run (`exchange_ab`)
    assert p["exchange_ab"] == ["factors_word_sharded", "sstats_allreduce", "factors_whole_stats"]
    for extra, want in ((["--whole-stats"], ("factors", False, False)),
                        (["--exchange", "sstats"], ("sstats", False, False)),
                        (["--exchange", "direct"], ("factors", True, True)),
                        (["--global-batch", "1600"], ("factors", True, False)),
                        (["--topics", "500", "--words", "100000", "--batch", "512"], ("factors", True, False)),
                        (["--topics", "10", "--words", "100", "--batch", "4000"], ("sstats", False, False))):
        d = tmp_path / ("x" + "_".join(extra).replace("-", ""))
        d.mkdir()
        r = run(8, {"TRLDA_BENCH_DRY_DIR": str(d)}, args=extra)
        assert r.returncode == 0, r.stderr
        ps = [json.load(open(d / ("plan%d.json" % i))) for i in range(8)]
        assert all(q == ps[0] for q in ps)
        assert (ps[0]["exchange"], ps[0]["word_sharded_m_step"], ps[0]["direct_requested"]) == want, (extra, ps[0])
