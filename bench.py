#!/usr/bin/env python3
"""bench.py -- E-step docs/sec (mini-batch) at K=100, V=7000 on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): OnlineLDA K=100, V=7000, mini-batches of
200 synthetic documents per GPU (Zipf-1.07 vocabulary, ~100 unique words per document),
max_iter_inference=20, threshold=1e-3.  A *step* is one full E-step of the reference's
LDA::updateVariablesVI (lda.cpp:160-220) over one mini-batch per GPU: row sums + exp E[log
beta] preamble, per-document gamma fixed point, sufficient statistics -- and, for N > 1, the
RCCL all-reduce of the K x V statistics (the path's one exchange step).  Inputs (lambda,
CSR batches and their word-major index, gamma0) are resident in HBM before the timed region;
nothing is cached across steps (lambda's preamble is recomputed every step, as in the
reference).  Weak scaling by default: 200 documents per GPU per step; `--global-batch 1600`
splits a fixed 1600-document step over the GPUs instead (BASELINE.json configs[2], "strong").

N = 1: every step announces the batch of the next step (trlda_model_estep_io_next), whose preamble
is then prepared by extra workgroups of this step's document-kernel launch, on the CUs that a
200-document batch leaves idle: one launch fewer per step, every step's preamble still computed
from lambda for its own words (`--no-prefetch`: a launch of its own).

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel
(estep_docs_kernel) from HIP events on the launch stream; `cpu_baseline` is the reference's
own C++ core (oracle/_ref, kind "reference") or, when that was not built, the plain-C port
(oracle/cpu_ref.c) timed on this box's host cores on a bounded sample of the same batches.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
SETTLE_MIN_S, SETTLE_MAX_S = 0.03, 0.3   # the untimed settle phase in front of the first timed region
FP64_PEAK_TFLOPS = 78.6        # fp64 vector: half the 157.3 TF fp32 vector peak of the same guide
KERNEL_NAMES = ["rowsum_partial_kernel", "exp_elog_beta_kernel", "estep_docs_kernel",
                "sstats_words_kernel"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--topics", type=int, default=100)
    ap.add_argument("--words", type=int, default=7000)
    ap.add_argument("--batch", type=int, default=200, help="documents per GPU per step")
    ap.add_argument("--max-iter", type=int, default=20)
    ap.add_argument("--threshold", type=float, default=1e-3)
    ap.add_argument("--mean-unique", type=int, default=100)
    ap.add_argument("--uniform", action="store_true", help="uniform instead of Zipf vocabulary")
    ap.add_argument("--lengths", choices=["poisson", "lognormal"], default="poisson",
                    help="unique words per document: 1 + Poisson(mean - 1) (SURVEY.md 8d, the "
                         "headline) or heavy-tailed: log-normal around the mean, ~2 %% of the "
                         "documents over 192 words, a few of 300..600 (utils/synthetic.py)")
    ap.add_argument("--num-batches", type=int, default=0,
                    help="distinct mini-batches streamed through the timed steps, one after the other "
                         "(0 = auto: 40 000 documents at N = 1 -- 200 batches of 200, SURVEY.md 8(d) config "
                         "2 -- but at least 8 batches; 8 per rank where every rank also holds the other "
                         "ranks' word lists)")
    ap.add_argument("--sstats-mode", choices=["segmented", "atomic"], default="segmented")
    ap.add_argument("--doc-threads", type=int, default=0)
    ap.add_argument("--split-preamble", action="store_true",
                    help="row sums and exp E[log beta] as two kernels even where one fused launch would do")
    ap.add_argument("--dense-preamble", action="store_true",
                    help="exp E[log beta] for all V words (reference behaviour) instead of the "
                         "batch's active words")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-only", action="store_true",
                    help="keep the parity leg (GPU against the oracle on the first timed batch) but "
                         "skip timing the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: this many documents per step in all, split over the "
                         "GPUs (BASELINE.json configs[2]: 1600); default 0 = weak scaling, "
                         "--batch documents per GPU")
    ap.add_argument("--exchange", choices=["auto", "factors", "sstats", "direct"], default="auto",
                    help="N > 1: what crosses ranks per step -- 'factors': an all-gather of every "
                         "document's expElogtheta row and per-entry weights (8 (K + n_d) bytes per "
                         "document, trlda_model_estep_dp), each rank then forms the statistics of the "
                         "whole mini-batch and the M-step in one kernel; 'sstats': the all-reduce of "
                         "K x V statistics; 'auto': whichever moves fewer bytes; 'direct': the "
                         "factors, written straight into the peers' buffers through hipIpc-mapped "
                         "pointers with a step counter instead of ncclAllGather "
                         "(trlda_model_dp_direct_*)")
    ap.add_argument("--virtual-world", type=int, default=0,
                    help="development aid for a one-GPU box: time what ONE rank of this many executes "
                         "per step with --exchange factors (the other ranks' slots hold a copy of "
                         "this rank's factors; no collective runs)")
    ap.add_argument("--whole-stats", action="store_true",
                    help="N > 1 with the factor exchange: every rank forms the statistics of the whole "
                         "mini-batch (rounds 2-3) instead of its range of the vocabulary + an exchange of "
                         "the lambda columns (word-sharded M-step, the default)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="every step launches its own preamble kernel instead of having it prepared "
                         "by extra workgroups of the previous step's document-kernel launch")
    ap.add_argument("--no-update-rates", action="store_true",
                    help="skip the secondary update_parameters figures (N = 1 only)")
    ap.add_argument("--batch-lda", action="store_true",
                    help="N > 1: close every step with BatchLDA's M-step, lambda = eta + sstats "
                         "(batchlda.cpp:60; BASELINE.json configs[3] with --topics 200 --words 50000 "
                         "--batch 12500 --max-iter 100), instead of OnlineLDA's blend with rho = 0.01")
    ap.add_argument("--headline-only", action="store_true",
                    help="profiling runs: only the headline steps (no second pass without the "
                         "prefetch), so that per-kernel means describe one kind of launch")
    ap.add_argument("--repeats", type=int, default=7,
                    help="the timed region (--steps steps) is run this many times back to back; "
                         "the median is reported, min / max beside it")
    ap.add_argument("--no-deferred", action="store_true",
                    help="every step launches its own statistics kernel (round 4's step) instead of "
                         "leaving them to the next step's document launch")
    ap.add_argument("--lanes", type=int, default=2, choices=(1, 2),
                    help="N = 1, deferred statistics: E-steps of the stream in flight at once "
                         "(trlda_model_set_stream_lanes: the steps go in turn to two streams of the "
                         "library's own, output arrays alternate between two sets; every fence() joins "
                         "them before the clock stops).  1 = one launch at a time, as in rounds 1-4")
    ap.add_argument("--no-exchange-ab", action="store_true",
                    help="N > 1: do not time the other exchange plans after the run's own")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the leg that starts every mini-batch from CSR arrays in host memory")
    ap.add_argument("--no-settle", action="store_true",
                    help="skip the untimed settle phase in front of the first timed region (clock ramp "
                         "after the set-up; reported as settle_steps / settle_ms)")
    ap.add_argument("--launch-timeout", type=float, default=1500.,
                    help="`--gpus N` without a launcher starts the N ranks itself; seconds after "
                         "which it ends them and fails")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="exercise the rank launch only (no torch, no GPU): CPU test of the path")
    return ap.parse_args()


def algorithmic_bytes(K, V, indptr):
    """bytes_alg = 32 K V + sum_d (16 K n_d + 8 n_d + 16 K) per E-step call,
    and the share of it that the document kernel itself moves (DESIGN.md 'Kernels')."""
    n = np.diff(indptr).astype(np.float64)
    B = len(n)
    per_doc = float((16. * K * n + 8. * n + 16. * K).sum())
    estep = 32. * K * V + per_doc
    # document kernel: gather beta_d once (8 K n), ids + counts (8 n), gamma in/out (16 K),
    # exp(psi(gamma)) out (8 K) and the per-word weights handed to the sstats kernel (8 n)
    docs_kernel = float((8. * K * n + 16. * n + 24. * K).sum())
    return estep, docs_kernel, B


def exchange_plan(args, world, virtual_world=0):
    """What crosses ranks in the N > 1 step, decided from the ARGUMENTS only, so
    that every rank takes the same decision without talking to the others: the factor exchange
    (all-gather of the documents' exp(psi(gamma)) rows and entry weights) when it moves fewer bytes
    than the all-reduce of the K x V statistics, with the statistics + M-step sharded by vocabulary
    range unless --whole-stats; `direct` (hipIpc pushes) only on request.  The run-time checks may
    fall back from it -- all ranks together -- never pick something else."""
    K, V = args.topics, args.words
    B = args.global_batch // world if args.global_batch > 0 else args.batch
    xworld = virtual_world or world
    slot = B * K + int(B * args.mean_unique * 1.2)   # ~ max_r(docs) K + max_r(nnz)
    factors_bytes, sstats_bytes = 8. * xworld * slot, 8. * 2. * K * V
    exchange = args.exchange if args.exchange not in ("auto", "direct") else \
        ("factors" if factors_bytes < sstats_bytes or args.exchange == "direct" else "sstats")
    if virtual_world:
        exchange = "factors"
    sharded = bool(exchange == "factors" and not args.whole_stats)
    own = ("direct" if args.exchange == "direct" else "factors_word_sharded" if sharded
           else "factors_whole_stats" if exchange == "factors" else "sstats_allreduce")
    return {"exchange": exchange, "slot_doubles": slot, "batch_per_gpu": B,
            "factors_bytes_per_rank": factors_bytes, "allreduce_bytes_per_rank": sstats_bytes,
            "word_sharded_m_step": sharded,
            "direct_requested": args.exchange == "direct",
            # the plans timed in one run (`exchange_ab` of the result line): the run's own first, then
            # the others -- the same list on every rank
            "exchange_ab": [] if (virtual_world or getattr(args, "no_exchange_ab", False)) else
            [own] + [p for p in ("sstats_allreduce", "factors_whole_stats", "factors_word_sharded") if p != own]}


def launch_ranks(n, argv, timeout_s):
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves.

    The parent never touches the GPU (torch is not even imported here): the ranks are CHILD
    processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment -- what
    torch.distributed.run would give them -- and rank 0's JSON line is relayed as the parent's
    one line of stdout.  Any rank failing, or the deadline passing, ends every rank (each child
    leads a process group of its own, killed by exact pid) and the parent exits non-zero
    without a JSON line."""
    import signal
    import socket
    import subprocess
    import tempfile
    assert "torch" not in sys.modules, "the launching parent must stay off the GPU runtime"
    with socket.socket() as s:                        # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, outs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   TRLDA_BENCH_PARENT_HAD_TORCH=str(int("torch" in sys.modules)))
        out = tempfile.TemporaryFile(mode="w+")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out, stderr=None, start_new_session=True))

    def end_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except OSError:
                    pass
        t_end = time.time() + 5.
        for p in procs:
            try:
                p.wait(max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()

    deadline = time.time() + timeout_s
    failed = None
    try:
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = "rank %d exited with status %d" % (r, p.returncode)
                    break
            if failed is None and time.time() > deadline:
                failed = "no result after %.0f s (--launch-timeout)" % timeout_s
            if failed is None:
                time.sleep(0.05)
        for r, p in enumerate(procs):
            if failed is None and p.poll() not in (None, 0):
                failed = "rank %d exited with status %d" % (r, p.returncode)
    finally:
        end_all()
    lines = []
    for r, out in enumerate(outs):
        out.seek(0)
        text = out.read()
        out.close()
        if r == 0:
            lines = [l for l in text.splitlines() if l.strip()]
        elif text.strip():
            sys.stderr.write(text)                   # other ranks' stdout is not the result
    result = None
    for l in lines:
        try:
            j = json.loads(l)
        except ValueError:
            j = None
        if isinstance(j, dict) and "metric" in j:
            result = l
        else:
            sys.stderr.write(l + "\n")
    if failed is None and result is None:
        failed = "rank 0 printed no result line"
    if failed is not None:
        sys.stderr.write("bench.py --gpus %d: %s; all ranks ended, no result\n" % (n, failed))
        return 1
    print(result, flush=True)
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: be the launcher (child processes, before any GPU call)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if args.dry_run_launch:
        # CPU check of the launch path (tests/test_bench_launch.py): no torch, no GPU
        if os.environ.get("TRLDA_BENCH_DRY_FAIL_RANK") == str(rank):
            sys.exit(7)
        if os.environ.get("TRLDA_BENCH_DRY_HANG_RANK") == str(rank):
            time.sleep(3600)
        mark = os.environ.get("TRLDA_BENCH_DRY_DIR")
        plan = exchange_plan(args, world)
        if mark:
            with open(os.path.join(mark, "rank%d" % rank), "w") as f:
                f.write("%d %d %s %s\n" % (rank, world, os.environ.get("MASTER_ADDR"),
                                          os.environ.get("TRLDA_BENCH_PARENT_HAD_TORCH")))
            with open(os.path.join(mark, "plan%d.json" % rank), "w") as f:
                json.dump(plan, f, sort_keys=True)
        if rank == 0:
            print(json.dumps({"metric": "dry run of the rank launch", "n_gpus": world,
                              "torch_in_rank": "torch" in sys.modules, "plan": plan}), flush=True)
        return

    # Only the result line goes to stdout: everything else that writes to file descriptor 1 from
    # here on -- RCCL's version banner, a library's printf -- lands on stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    if torch.cuda.device_count() < world:             # (counting does not initialise the GPU)
        sys.stderr.write("bench.py: %d GPUs requested, %d visible on this node\n"
                         % (world, torch.cuda.device_count()))
        sys.exit(3)
    import torch.distributed as dist
    from trlda_amd import _ffi, build
    from trlda_amd.documents import CSRDocuments, DeviceBatch
    from trlda_amd.utils.synthetic import SEED_BASE, make_corpus

    if not os.path.exists(_ffi.LIB_PATH):
        build.build()
    L = _ffi.lib()
    _ffi.require_gpu()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # TRLDA_BENCH_FORCE_DIST=1: exercise the N > 1 code path (process group, all-reduce,
    # barrier) on a single GPU -- a development aid for boxes with one device
    force_dist = world == 1 and os.environ.get("TRLDA_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group("nccl", device_id=device)
    collective = world > 1 or force_dist

    # The exchange through the C ABI: a communicator of our own (ncclCommInitRank on the id that
    # rank 0 broadcasts), handed to trlda_model_allreduce_sstats, which enqueues ncclAllReduce on
    # the model's stream -- the same collective as torch.distributed.all_reduce without ~9 us of
    # framework per call (profiles/r02_allreduce_world1.txt).  Checked against torch's result
    # once; any failure falls back to torch.distributed.
    rccl_comm = None
    if collective and os.environ.get("TRLDA_BENCH_TORCH_ALLREDUCE") != "1":
        try:
            from trlda_amd import rccl
            rccl_comm = rccl.own_communicator(dist, device)
        except Exception as exc:                      # noqa: BLE001
            if rank == 0:
                print("bench: own RCCL communicator unavailable (%s); using torch.distributed" % exc,
                      file=sys.stderr)
            rccl_comm = None

    rccl_ranks = None
    if rccl_comm is not None:
        from trlda_amd import rccl
        rccl_ranks = rccl.comm_count(rccl_comm)       # what RCCL itself says, not WORLD_SIZE

    K, V = args.topics, args.words
    strong = args.global_batch > 0
    if strong:
        # documents [rank * B, (rank + 1) * B) of each global mini-batch: equal counts (the
        # synthetic documents are i.i.d., so the shares of nnz are equal to within a percent)
        if args.global_batch % world:
            raise SystemExit("--global-batch must be a multiple of the number of GPUs")
        args.batch = args.global_batch // world
    B = args.batch
    KV = K * V

    # ---- inputs, resident in HBM before the timed region ---------------------------------
    L.trlda_seed(1)                                   # libc stream: lambda0 then gamma0s
    lam = np.empty((K, V), order="F")
    L.trlda_sample_gamma_init(K, V, lam)              # replicated lambda (same seed per rank)
    model = _ffi.vp()
    _ffi.check(L.trlda_model_create(C.byref(model), local_rank, K, V))
    stream = torch.cuda.current_stream(device).cuda_stream
    _ffi.check(L.trlda_model_set_stream(model, _ffi.vp(stream)))
    _ffi.check(L.trlda_model_set_lambda(model, lam))
    _ffi.check(L.trlda_model_set_alpha(model, np.full(K, .1)))
    _ffi.check(L.trlda_model_set_sstats_mode(model, 1 if args.sstats_mode == "atomic" else 0))
    _ffi.check(L.trlda_model_set_doc_threads(model, args.doc_threads))
    _ffi.check(L.trlda_model_set_dense_preamble(model, int(args.dense_preamble)))
    _ffi.check(L.trlda_model_set_split_preamble(model, int(args.split_preamble)))
    _ffi.check(L.trlda_model_set_word_sharding(model, int(not args.whole_stats)))

    if collective:
        # every rank has the communicator, or none uses it (a rank on its own in a collective hangs)
        have = torch.tensor([int(rccl_comm is not None)], device=device)
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        if int(have.item()) != 1:
            rccl_comm = None
    if rccl_comm is not None:                         # one check against torch's collective
        probe = torch.ones(KV, dtype=torch.float64, device=device) * (rank + 1)
        want = probe.clone()
        dist.all_reduce(want)
        rc = L.trlda_model_allreduce_sstats(model, rccl_comm, C.c_void_p(probe.data_ptr()))
        torch.cuda.synchronize()
        ok = torch.tensor([int(rc == 0 and bool(torch.equal(probe, want)))], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            rccl_comm = None                          # every rank falls back together
        del probe, want

    # ---- what crosses ranks -------------------------------------------------
    vworld = args.virtual_world if (args.virtual_world > 1 and world == 1) else 0
    if vworld:
        collective = True                            # the M-step inside the step, no prefetch
    xworld = vworld or world                         # ranks the mini-batch is cut over
    def rank_corpus(r, i):
        seed = SEED_BASE + 1 + 1000 * r + i          # config index 1; distinct per rank
        lengths = None
        if args.lengths == "lognormal":
            from trlda_amd.utils.synthetic import lognormal_lengths
            lengths = lognormal_lengths(B, seed, median=args.mean_unique, longest=min(600, V))
        return CSRDocuments(*make_corpus(B, V, seed=seed, mean_unique=args.mean_unique,
                                         zipf=not args.uniform, lengths=lengths))
    exchange, direct = "none", False
    if collective:
        # from the arguments only: the same choice on every rank (exchange_plan, above)
        plan = exchange_plan(args, world, vworld)
        slot, exchange = plan["slot_doubles"], plan["exchange"]
        if args.exchange == "direct" and world > 1:
            # every rank exports its region, the handles travel through the process group,
            # every rank maps its peers' regions (all ranks together, or none)
            mine = C.create_string_buffer(64)
            ok = L.trlda_model_dp_direct_alloc(model, 2 * slot, world, mine) == 0
            t = torch.frombuffer(bytearray(mine.raw + bytes([int(ok)])), dtype=torch.uint8).clone().to(device)
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            raw = [p_.cpu().numpy().tobytes() for p_ in parts]
            ok = all(r_[64] == 1 for r_ in raw) and \
                L.trlda_model_dp_direct_connect(model, rank, world, b"".join(r_[:64] for r_ in raw)) == 0
            flag = torch.tensor([int(ok)], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            direct = int(flag.item()) == 1
            if not direct:
                L.trlda_model_dp_direct_close(model)
        if exchange == "factors" and world > 1 and rccl_comm is None and not direct:
            exchange = "sstats"                      # ncclAllGather needs the communicator

    if args.num_batches <= 0:
        # 40 000 documents streamed (200 batches of 200), at least 8 batches
        args.num_batches = max(8, min(200, 40000 // max(B, 1))) if (world == 1 and not vworld and
                                                                    not force_dist) else 8
    batches, csrs, gamma0s, gbatches = [], [], [], []
    cuts = (np.arange(xworld + 1) * B).astype(np.int32)
    # (batch i is the same documents whatever --num-batches is: a seed per batch; NumPy's samplers
    # release the GIL, so the corpus is generated on a few host threads)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 2) // 2))) as pool:
        own_csrs = list(pool.map(lambda i: rank_corpus(rank, i), range(args.num_batches)))
    for i in range(args.num_batches):
        csr = own_csrs[i]
        csrs.append(csr)
        batches.append(DeviceBatch(csr, V, local_rank))
        if exchange == "factors":
            # every rank holds the whole mini-batch (word lists only: a few hundred kB)
            parts = [csr if r == rank else rank_corpus(r, i) for r in range(xworld)]
            off = np.concatenate([[0], np.cumsum([int(p.indptr[-1]) for p in parts])])
            whole = CSRDocuments(
                np.concatenate([parts[0].indptr] + [p.indptr[1:] + off[r] for r, p in enumerate(parts) if r]),
                np.concatenate([p.ids for p in parts]), np.concatenate([p.cnts for p in parts]))
            gbatches.append(DeviceBatch(whole, V, local_rank))
        g0 = np.empty((K, B), order="F")
        L.trlda_sample_gamma_init(K, B, g0)
        gamma0s.append(torch.from_numpy(np.ascontiguousarray(g0.T)).to(device))
    gamma = torch.empty(B * K, dtype=torch.float64, device=device)
    sstats = torch.empty(KV, dtype=torch.float64, device=device)
    iters_dev = torch.zeros(B, dtype=torch.int32, device=device)
    # (two E-steps in flight write two sets of arrays: --lanes)
    out_sets = [(gamma, sstats, iters_dev),
                (torch.empty_like(gamma), torch.empty_like(sstats), torch.zeros_like(iters_dev))]
    # N > 1: the M-step (onlinelda.cpp:99-100) closes the dependency chain E-step -> all-reduce
    # -> lambda -> next E-step, so the collective cannot hide behind the next step's kernels.
    # lambda' stays the initial lambda and rho is small: lambda moves, the workload does not drift.
    lam_prime = torch.from_numpy(np.ascontiguousarray(lam.ravel(order="F"))).to(device) \
        if collective else None
    RHO, ETA, D_TOTAL = 0.01, 0.3, 1000000
    if args.batch_lda:
        # BatchLDA's M-step (batchlda.cpp:60): lambda = eta + sstats -- rho = 1, no D / B scaling;
        # the steps are then consecutive epochs-on-a-mini-batch of a real BatchLDA run
        RHO, D_TOTAL = 1.0, None

    prefetch = not collective and not args.no_prefetch
    # Deferred statistics (include/trlda_hip.h, trlda_model_set_deferred_stats): the headline is a
    # stream of E-steps on a fixed lambda; the statistics of step i ride on step i + 1's document
    # launch.  Every fence() flushes what is pending BEFORE the clock stops, so a timed region of
    # K steps holds K statistics stages: K - 1 inside document launches + the last one as a kernel.
    deferred = prefetch and not args.no_deferred
    _ffi.check(L.trlda_model_set_deferred_stats(model, int(deferred)))
    # Stream lanes (include/trlda_hip.h, trlda_model_set_stream_lanes): the steps of that stream
    # share nothing but lambda, so two of them may be in flight -- step i + 1's workgroups take the
    # CUs as step i's leave them (its documents end 1-2 us apart, a 129..144-word one 5 us after
    # the others).  Every step still is one E-step on one mini-batch, its results bitwise those
    # of the one-lane stream; fence() joins the lanes before the clock stops.
    lanes = [args.lanes if deferred else 1]
    _ffi.check(L.trlda_model_set_stream_lanes(model, lanes[0]))
    upcoming = (C.c_void_p * 2)()
    exchange_probe = None
    if exchange == "factors" and not vworld and (world > 1 or force_dist):
        # One check of the factor path against the trusted composition (E-step on the shard,
        # torch.distributed's all-reduce of the K x V statistics) on the first mini-batch: every
        # rank must see the whole mini-batch's statistics.  Any failure, on any rank, sends all
        # ranks to the all-reduce path together.
        ok = 0
        try:
            want = torch.empty(KV, dtype=torch.float64, device=device)
            got = torch.empty(KV, dtype=torch.float64, device=device)
            _ffi.check(L.trlda_model_estep_io(model, batches[0].handle, gamma0s[0].data_ptr(),
                                              gamma.data_ptr(), want.data_ptr(), args.max_iter,
                                              args.threshold, None))
            torch.cuda.synchronize()
            dist.all_reduce(want)
            _ffi.check(L.trlda_model_estep_dp(
                model, gbatches[0].handle, batches[0].handle, rccl_comm, rank, world,
                cuts.ctypes.data_as(C.POINTER(C.c_int32)), gamma0s[0].data_ptr(), gamma.data_ptr(),
                got.data_ptr(), args.max_iter, args.threshold, None, 0, None, 0., 0., 0.))
            torch.cuda.synchronize()
            scale_ = float(want.abs().max().item())
            diff = float((got - want).abs().max().item())
            ok = int(scale_ > 0 and diff <= 1e-9 * scale_)
            exchange_probe = {"max_abs_diff_vs_allreduce": diff, "max_abs": scale_}
            del want, got
        except Exception as exc:                      # noqa: BLE001
            exchange_probe = {"error": repr(exc)[:200]}
        flag = torch.tensor([ok], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            exchange = "sstats"
            if rank == 0:
                print("bench: factor exchange failed its check (%s); using the all-reduce of sstats"
                      % (exchange_probe,), file=sys.stderr)
    if exchange == "factors" and not vworld and not args.whole_stats and \
            (world > 1 or (force_dist and os.environ.get("TRLDA_BENCH_SHARD_CHECK") == "1")):
        # One check of the word-sharded M-step (grouped ncclBroadcast of the ranks' lambda ranges,
        # never run on this box's one GPU): a step on the first mini-batch must leave the SAME
        # lambda on every rank, and the one that the whole-statistics form leaves.  Any failure, on
        # any rank, switches the sharding off on all ranks together.
        ok, detail = 0, None
        try:
            def one_step(sharded):
                _ffi.check(L.trlda_model_set_lambda(model, lam))
                _ffi.check(L.trlda_model_set_word_sharding(model, int(sharded)))
                _ffi.check(L.trlda_model_estep_dp(
                    model, gbatches[0].handle, batches[0].handle, rccl_comm, rank, world,
                    cuts.ctypes.data_as(C.POINTER(C.c_int32)), gamma0s[0].data_ptr(), gamma.data_ptr(),
                    None, args.max_iter, args.threshold, None, 1, lam_prime.data_ptr(), RHO, ETA,
                    D_TOTAL / float(B * world) if D_TOTAL else 1.))
                got = np.empty((K, V), order="F")
                _ffi.check(L.trlda_model_get_lambda(model, got))
                return got, bool(L.trlda_model_last_word_sharded(model))
            lam_sh, was_sharded = one_step(True)
            lam_wh, _ = one_step(False)
            chk = torch.tensor([float(lam_sh.sum()), -float(lam_sh.sum())], dtype=torch.float64, device=device)
            dist.all_reduce(chk, op=dist.ReduceOp.MAX)
            same = float(chk[0].item()) == -float(chk[1].item())
            dev_ = float(np.max(np.abs(lam_sh - lam_wh) / lam_wh))
            ok = int(was_sharded and same and dev_ < 1e-9)
            detail = {"word_sharded": was_sharded, "replicas_equal": same, "max_rel_dev_vs_whole_statistics": dev_}
        except Exception as exc:                      # noqa: BLE001
            detail = {"error": repr(exc)[:200]}
        flag = torch.tensor([ok], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        sharding_ok = int(flag.item()) == 1
        _ffi.check(L.trlda_model_set_word_sharding(model, int(sharding_ok)))
        _ffi.check(L.trlda_model_set_lambda(model, lam))
        exchange_probe = dict(exchange_probe or {}, word_sharded_m_step=detail)
        if not sharding_ok and rank == 0:
            print("bench: word-sharded M-step failed its check (%s); every rank forms the whole "
                  "mini-batch's statistics" % (detail,), file=sys.stderr)
    if vworld:
        # no collective: the first call copies this rank's slot into the other ranks' (finite,
        # plausible factors), later calls leave the buffer alone
        HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
        filled = []

        def fill_once(ctx, send, recv, count, hip_stream):
            if not filled:
                filled.append(1)
                torch.cuda.synchronize()
                own = np.empty(int(count))
                L.trlda_dev_download(local_rank, C.c_void_p(own.ctypes.data), C.c_void_p(send), own.nbytes)
                for r in range(1, vworld):
                    L.trlda_dev_upload(local_rank, C.c_void_p(recv + r * own.nbytes),
                                       C.c_void_p(own.ctypes.data), own.nbytes)
            return 0
        virtual_hook = HOOK(fill_once)
        _ffi.check(L.trlda_model_set_allgather(model, C.cast(virtual_hook, C.c_void_p), None))
        # the word-sharded M-step's exchange of lambda columns: nothing moves (the other ranks'
        # ranges keep their old values: finite, plausible); --whole-stats: every rank forms the
        # whole mini-batch's statistics as in rounds 2-3
        HOOKV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_int, C.c_int,
                            C.c_void_p)
        virtual_hookv = HOOKV(lambda ctx, table, offs, r, w, st: 0)
        if not args.whole_stats:
            _ffi.check(L.trlda_model_set_allgatherv(model, C.cast(virtual_hookv, C.c_void_p), None))

    cuts_solo = np.array([0, B], dtype=np.int32)
    use_prefetch = [prefetch]

    def step(i, want_iters=False, plain=False, solo=False, threshold=None, out_set=None):
        thr = args.threshold if threshold is None else threshold
        # (whose results are read afterwards: set 0; a stream through two lanes: alternating)
        g_out, s_out, it_out = out_sets[out_set if out_set is not None else
                                        (i & 1 if lanes[0] > 1 and not want_iters else 0)]
        # plain: the bare E-step (parity leg).  solo: the N > 1 step without its exchange -- this
        # rank's documents, the same kernels, the M-step -- run by one rank on its own
        j = i % args.num_batches
        if solo and exchange == "factors":
            _ffi.check(L.trlda_model_estep_dp(
                model, batches[j].handle, batches[j].handle, None, 0, 1,
                cuts_solo.ctypes.data_as(C.POINTER(C.c_int32)), gamma0s[j].data_ptr(), gamma.data_ptr(),
                None, args.max_iter, thr, None, 1, lam_prime.data_ptr(), RHO, ETA,
                D_TOTAL / float(B) if D_TOTAL else 1.))
            return
        # gamma0 is read-only input, gamma the output (lda.cpp:168 copies, we do not).  The batch
        # of the next step is announced: its preamble (row sums + exp(psi(lambda)) on ITS words,
        # recomputed for every step) is prepared by extra workgroups of this step's document-kernel
        # launch, on the CUs a 200-document batch leaves idle -- one launch fewer per step,
        # nothing skipped, nothing shared between steps.  N > 1: the M-step changes lambda every
        # step, so there is nothing to prepare ahead.
        if exchange == "factors" and not plain:
            # documents of this rank -> all-gather of the factors -> statistics of the whole
            # mini-batch + M-step (onlinelda.cpp:99-100) in one kernel, on every rank
            _ffi.check(L.trlda_model_estep_dp(
                model, gbatches[j].handle, batches[j].handle, rccl_comm, 0 if vworld else rank, xworld,
                cuts.ctypes.data_as(C.POINTER(C.c_int32)), gamma0s[j].data_ptr(), gamma.data_ptr(),
                None, args.max_iter, thr, iters_dev.data_ptr() if want_iters else None, 1,
                lam_prime.data_ptr(), RHO, ETA, D_TOTAL / float(B * xworld) if D_TOTAL else 1.))
            return
        if use_prefetch[0]:
            # the batches of the next two steps, in order: with two lanes a step's launch prepares the
            # preamble of the step AFTER the next (one lane: of the next, as trlda_model_estep_io_next)
            upcoming[0] = batches[(i + 1) % args.num_batches].handle.value
            upcoming[1] = batches[(i + 2) % args.num_batches].handle.value
            _ffi.check(L.trlda_model_estep_io_ahead(model, batches[j].handle, upcoming, 2,
                                                    gamma0s[j].data_ptr(), g_out.data_ptr(), s_out.data_ptr(),
                                                    args.max_iter, thr,
                                                    it_out.data_ptr() if want_iters else None))
        else:
            _ffi.check(L.trlda_model_estep_io_next(model, batches[j].handle, None, gamma0s[j].data_ptr(),
                                                   g_out.data_ptr(), s_out.data_ptr(), args.max_iter, thr,
                                                   it_out.data_ptr() if want_iters else None))
        if collective and not plain:
            if solo:
                pass                                  # the sum over one rank
            elif rccl_comm is not None:               # RCCL over xGMI: K x V fp64 sum
                _ffi.check(L.trlda_model_allreduce_sstats(model, rccl_comm,
                                                          C.c_void_p(s_out.data_ptr())))
            else:
                dist.all_reduce(s_out)
            _ffi.check(L.trlda_model_blend(model, lam_prime.data_ptr(), s_out.data_ptr(), RHO, ETA,
                                           D_TOTAL / float(B * (1 if solo else world)) if D_TOTAL else 1.))

    def fence():
        _ffi.check(L.trlda_model_flush(model))       # deferred statistics: enqueued now, waited for below
        if collective and not vworld:
            dist.barrier()
        torch.cuda.synchronize()

    pos = [args.warmup]                              # the stream of mini-batches goes on across repeats

    def timed(**kw):
        """EXACTLY args.steps steps between two fences; the maximum over ranks.  Every timed
        region takes up the stream of mini-batches where the last one stopped (each step announces
        its successor: nothing is announced twice, nothing comes unannounced)."""
        fence()
        first = pos[0]
        pos[0] += args.steps
        t0 = time.perf_counter()
        for i in range(first, first + args.steps):
            step(i, **kw)
        fence()
        dt = time.perf_counter() - t0
        if collective and not vworld:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for i in range(args.warmup):
        step(i)

    def settle(**kw):
        """An UNTIMED, declared settle phase in front of a timed leg: `--warmup
        5` is 0.2 ms of device work after seconds of host-side set-up, and the first timed region
        then runs on a GPU whose clocks are still on their way up -- in round 4's driver run the
        same fixed work measured 43.4 us per step in the first leg and 39.9 two legs later.  Steps
        of the same stream of mini-batches, in samples of min(--steps, 20), until two consecutive
        samples agree within 2 % and at least SETTLE_MIN_S have passed, at most SETTLE_MAX_S; the
        decision is taken on the maximum over ranks, so every rank runs the same number of steps.
        `warmup` in the result line stays what was asked for; this phase is reported beside it
        (`settle_steps`, `settle_ms`) and `--no-settle` switches it off."""
        if args.no_settle:
            return {"settle_steps": 0, "settle_ms": 0.0}
        chunk, total, n, last = max(1, min(args.steps, 20)), 0.0, 0, None
        while True:
            fence()
            first = pos[0]
            pos[0] += chunk
            t0 = time.perf_counter()
            for i in range(first, first + chunk):
                step(i, **kw)
            fence()
            dt = time.perf_counter() - t0
            if collective and not vworld:
                t = torch.tensor([dt], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            total += dt
            n += chunk
            agree = last is not None and abs(dt - last) <= 0.02 * max(dt, last)
            last = dt
            if (agree and total >= SETTLE_MIN_S) or total >= SETTLE_MAX_S:
                break
        return {"settle_steps": n, "settle_ms": round(1e3 * total, 3),
                "settle_last_ms_per_step": round(1e3 * last / chunk, 5)}

    settled = settle()
    def lane_report():
        two, one = C.c_double(), C.c_double()
        L.trlda_model_lane_timing(model, C.byref(two), C.byref(one))
        return int(L.trlda_model_lane_state(model)), {"launch_us": round(two.value, 3), "step_us": round(one.value, 3),
                                                      "launches_in_flight": round(two.value / one.value, 3) if one.value else None}

    # the timed region, `--repeats` times back to back: the median is the result (20 steps of
    # 40 us are under a millisecond -- one scheduling hiccup moves a single sample by 10 %)
    samples = sorted(timed() for _ in range(max(1, args.repeats)))
    # (did the last launch of a timed region carry the statistics of an earlier step? -- asked here:
    # the first launches after a fence never do)
    carried_flag = bool(deferred and L.trlda_model_last_deferred(model) & 2)
    lane_state, lane_cal = lane_report()
    elapsed = samples[len(samples) // 2]
    docs_per_s = world * B * args.steps / elapsed
    repeats = {"n": len(samples), "value": "median",
               "ms_per_step_min": round(1e3 * samples[0] / args.steps, 5),
               "ms_per_step_max": round(1e3 * samples[-1] / args.steps, 5)}

    # N = 1: the same steps with every preamble in a launch of its own (what a caller gets who
    # does not announce its next batch, e.g. OnlineLDA.do_e_step)
    value_no_prefetch = None
    if prefetch and not args.headline_only:
        use_prefetch[0] = False
        for i in range(min(args.warmup, 5)):
            step(pos[0] + i)
        pos[0] += min(args.warmup, 5)
        s_np = sorted(timed() for _ in range(max(1, args.repeats)))
        value_no_prefetch = {"value": round(B * args.steps / s_np[len(s_np) // 2], 1), "unit": "docs/s",
                             "ms_per_step": round(1e3 * s_np[len(s_np) // 2] / args.steps, 5)}
        use_prefetch[0] = True
        for i in range(2):
            step(pos[0] + i)
        pos[0] += 2

    # the same steps one launch at a time (rounds 1-4's stream: what rocprofv3's per-launch durations
    # of `--lanes 1` describe)
    value_one_lane = None
    if lanes[0] > 1 and not args.headline_only:
        fence()
        lanes[0] = 1
        _ffi.check(L.trlda_model_set_stream_lanes(model, 1))
        for i in range(min(args.warmup, 5)):
            step(pos[0] + i)
        pos[0] += min(args.warmup, 5)
        s_1l = sorted(timed() for _ in range(max(1, args.repeats)))
        value_one_lane = {"value": round(B * args.steps / s_1l[len(s_1l) // 2], 1), "unit": "docs/s",
                          "ms_per_step": round(1e3 * s_1l[len(s_1l) // 2] / args.steps, 5)}
        fence()
        lanes[0] = args.lanes
        _ffi.check(L.trlda_model_set_stream_lanes(model, lanes[0]))
        for i in range(4):
            step(pos[0] + i)
        pos[0] += 4

    # the same steps with threshold = 0: FIXED work, every document runs all max_iter iterations
    # whatever lambda is ("threshold 1e-3 and threshold 0")
    value_fixed_work = None
    if not collective and not args.headline_only:
        for i in range(min(args.warmup, 5)):
            step(pos[0] + i, threshold=0.)
        pos[0] += min(args.warmup, 5)
        s_fw = sorted(timed(threshold=0.) for _ in range(max(1, args.repeats)))
        value_fixed_work = {"value": round(B * args.steps / s_fw[len(s_fw) // 2], 1), "unit": "docs/s",
                            "ms_per_step": round(1e3 * s_fw[len(s_fw) // 2] / args.steps, 5),
                            "threshold": 0.0, "iterations_per_document": args.max_iter}
        for i in range(2):
            step(pos[0] + i)
        pos[0] += 2

    # N = 1: the same stream of steps with NOTHING pre-uploaded: every mini-batch
    # starts as CSR arrays in host memory, goes through trlda_batch_create (validation + a copy on this
    # thread; the word-major index on the library's worker threads; its upload enqueued by this thread
    # again, in a later call) eight steps before its E-step, and is destroyed four steps after it.  PCIe and host work inclusive: reported beside
    # `value`, never as it.
    value_end_to_end = None
    if not collective and prefetch and not args.headline_only and not args.no_end_to_end:
        from trlda_amd.documents import DeviceBatch
        fence()
        nb = args.num_batches
        window = {}
        AHEAD = 8

        def e2e_run(n_steps):
            first = pos[0]
            pos[0] += n_steps
            for i in range(first, first + AHEAD):                    # the first ones of the stretch
                if i not in window:
                    window[i] = DeviceBatch(csrs[i % nb], V, local_rank)
            t0 = time.perf_counter()
            for i in range(first, first + n_steps):
                # (made AHEAD steps before its E-step, announced two steps before it: an index is there
                # ~170 us after its trlda_batch_create -- an announced batch whose index is not there yet
                # counts as not announced)
                window[i + AHEAD] = DeviceBatch(csrs[(i + AHEAD) % nb], V, local_rank)
                upcoming[0] = window[i + 1].handle.value
                upcoming[1] = window[i + 2].handle.value
                g_out, s_out, _ = out_sets[i & 1 if lanes[0] > 1 else 0]
                _ffi.check(L.trlda_model_estep_io_ahead(model, window[i].handle, upcoming, 2,
                                                        gamma0s[i % nb].data_ptr(), g_out.data_ptr(),
                                                        s_out.data_ptr(), args.max_iter, args.threshold, None))
                old = window.pop(i - 4, None)
                if old is not None:
                    old.close()
            fence()
            return time.perf_counter() - t0

        # (a stretch of at least 128 mini-batches: the stream ends with a fence -- two launches and the last
        # statistics drain with nothing behind them, ~100 us -- which a 20-step window would show as 5 us
        # per step; `mini_batches` in the line)
        n_e2e = max(args.steps, 128)
        e2e_run(max(n_e2e, 40))                                       # staging buffers, allocations, workers
        s_e2e = sorted(e2e_run(n_e2e) for _ in range(max(1, args.repeats)))
        for b_ in window.values():
            b_.close()
        window.clear()
        value_end_to_end = {"value": round(B * n_e2e / s_e2e[len(s_e2e) // 2], 1), "unit": "docs/s",
                            "mini_batches": n_e2e,
                            "ms_per_step": round(1e3 * s_e2e[len(s_e2e) // 2] / n_e2e, 5),
                            "ms_per_step_min": round(1e3 * s_e2e[0] / n_e2e, 5),
                            "what": "CSR arrays in host memory -> trlda_batch_create (index on the library's "
                                    "worker threads, TRLDA_INDEX_THREADS, default 4; uploads enqueued by the "
                                    "caller's thread) eight steps "
                                    "before its E-step, announced two steps before it -> E-step -> "
                                    "trlda_batch_destroy four steps after it; one Python thread drives it",
                            "host_threads": 1 + int(os.environ.get("TRLDA_INDEX_THREADS", "4"))}
        # ... and the same pass as ONE library call (trlda_model_estep_corpus: the loop over the
        # mini-batches inside the library, no Python between the steps): the corpus' CSR arrays in
        # host memory, gamma0 / gamma for all of it on the device, a ring of four statistics arrays
        try:
            n_c = min(nb, n_e2e)
            offs = np.zeros(n_c * B + 1, dtype=np.int64)
            at = 0
            for i in range(n_c):
                ip_ = csrs[i].indptr.astype(np.int64)
                offs[i * B:(i + 1) * B + 1] = ip_ + at
                at += int(ip_[-1])
            ids_c = np.concatenate([csrs[i].ids for i in range(n_c)])
            cnts_c = np.concatenate([csrs[i].cnts for i in range(n_c)])
            g0_c = torch.cat([gamma0s[i] for i in range(n_c)]).contiguous()
            g_c = torch.empty_like(g0_c)
            ring_t = [torch.empty(KV, dtype=torch.float64, device=device) for _ in range(4)]
            ring = (C.c_void_p * 4)(*[t_.data_ptr() for t_ in ring_t])

            def corpus_run():
                fence()
                t0 = time.perf_counter()
                _ffi.check(L.trlda_model_estep_corpus(model, n_c * B, offs.ctypes.data, ids_c.ctypes.data,
                                                      cnts_c.ctypes.data, B, g0_c.data_ptr(), g_c.data_ptr(), ring,
                                                      4, args.max_iter, args.threshold, None))
                fence()
                return time.perf_counter() - t0
            corpus_run()
            s_c = sorted(corpus_run() for _ in range(max(1, args.repeats)))
            value_end_to_end["one_call"] = {
                "value": round(B * n_c / s_c[len(s_c) // 2], 1), "unit": "docs/s", "mini_batches": n_c,
                "ms_per_step": round(1e3 * s_c[len(s_c) // 2] / n_c, 5),
                "ms_per_step_all": [round(1e3 * x / n_c, 5) for x in s_c],
                "lane_state_after": int(L.trlda_model_lane_state(model)),
                "what": "trlda_model_estep_corpus: the same pass over %d mini-batches from one CSR corpus in host "
                        "memory, the loop inside the library" % n_c}
            del g0_c, g_c, ring_t
            _ffi.check(L.trlda_model_set_deferred_stats(model, int(deferred)))
            _ffi.check(L.trlda_model_set_stream_lanes(model, lanes[0]))
        except Exception as exc:                      # noqa: BLE001
            value_end_to_end["one_call"] = {"error": repr(exc)[:200]}
        for i in range(4):
            step(pos[0] + i)
        pos[0] += 4
        # (the lanes are judged on the stream they carry: a verdict reached on THIS leg's stream -- host-paced,
        # uploads beside the launches -- is that leg's; the legs behind it time the resident stream again)
        if lane_state == 2 and int(L.trlda_model_lane_state(model)) == 1:
            fence()
            _ffi.check(L.trlda_model_set_stream(model, _ffi.vp(stream)))   # (its lanes are looked at afresh)
            value_end_to_end["lanes_given_up_on_this_leg"] = True
            for i in range(8):
                step(pos[0] + i)
            pos[0] += 8
            fence()

    # N > 1: the identical step (documents -> statistics -> M-step, no prefetch) WITHOUT the
    # exchange, timed on rank 0 alone while the other ranks wait: the like-for-like one-GPU
    # figure that the N-GPU value is to be divided by
    same_step_n1 = None
    if collective and not vworld:
        dist.barrier()
        if rank == 0:
            for i in range(min(args.warmup, 5)):
                step(i, solo=True)
            s_solo = []
            for _ in range(max(1, args.repeats)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    step(i, solo=True)
                torch.cuda.synchronize()
                s_solo.append(time.perf_counter() - t0)
            s_solo.sort()
            same_step_n1 = {"value": round(B * args.steps / s_solo[len(s_solo) // 2], 1),
                            "unit": "docs/s", "n_gpus": 1,
                            "ms_per_step": round(1e3 * s_solo[len(s_solo) // 2] / args.steps, 5),
                            "what": "rank 0 alone, same kernels and M-step, no exchange"}
        dist.barrier()
        _ffi.check(L.trlda_model_set_lambda(model, lam))   # every rank back on the common lambda

    # N > 1: the OTHER transports of the same step, a few steps each, in the same run (VERDICT r5 item 8:
    # the first lease of an 8-GPU node should answer "which exchange" at once): the all-reduce of the
    # K x V statistics | the factor exchange with every rank forming the whole mini-batch's statistics |
    # the factor exchange with the M-step sharded by vocabulary range (+ the lambda columns exchanged).
    # Decided from the arguments and the communicator alone, so every rank walks the same list; the
    # direct pushes are timed only when they are the run's own plan (--exchange direct: their regions
    # are mapped at start-up).  Reported, never `value`.
    exchange_ab = None
    if collective and not vworld and not args.no_exchange_ab:
        exchange_ab = {}
        word_sharded = bool(exchange == "factors" and L.trlda_model_last_word_sharded(model))
        plan_now = ("direct" if direct else "factors_word_sharded" if (exchange == "factors" and word_sharded)
                    else "factors_whole_stats" if exchange == "factors" else "sstats_allreduce")
        exchange_ab[plan_now] = round(1e6 * elapsed / args.steps, 3)
        keep_exchange, keep_sharding = exchange, int(bool(word_sharded))
        can_factors = rccl_comm is not None and not direct
        for name, ex, sh in (("sstats_allreduce", "sstats", None), ("factors_whole_stats", "factors", 0),
                             ("factors_word_sharded", "factors", 1)):
            if name == plan_now:
                continue
            if ex == "factors" and not can_factors:
                exchange_ab[name] = None             # (ncclAllGather needs the library's own communicator)
                continue
            try:
                if ex == "factors" and not gbatches:
                    for i in range(args.num_batches):    # every rank holds the whole mini-batch's word lists
                        parts = [csrs[i] if r == rank else rank_corpus(r, i) for r in range(xworld)]
                        off = np.concatenate([[0], np.cumsum([int(p_.indptr[-1]) for p_ in parts])])
                        whole = CSRDocuments(
                            np.concatenate([parts[0].indptr] + [p_.indptr[1:] + off[r] for r, p_ in enumerate(parts) if r]),
                            np.concatenate([p_.ids for p_ in parts]), np.concatenate([p_.cnts for p_ in parts]))
                        gbatches.append(DeviceBatch(whole, V, local_rank))
                exchange = ex
                if sh is not None:
                    _ffi.check(L.trlda_model_set_word_sharding(model, sh))
                _ffi.check(L.trlda_model_set_lambda(model, lam))
                n_ab = max(4, min(args.steps, 20))
                for i in range(3):
                    step(i)
                fence()
                t0 = time.perf_counter()
                for i in range(n_ab):
                    step(3 + i)
                fence()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                exchange_ab[name] = round(1e6 * float(t.item()) / n_ab, 3)
            except Exception as exc:                  # noqa: BLE001 (a transport that cannot run here)
                exchange_ab[name] = "failed: " + repr(exc)[:120]
        exchange = keep_exchange
        _ffi.check(L.trlda_model_set_word_sharding(model, keep_sharding))
        _ffi.check(L.trlda_model_set_lambda(model, lam))
        exchange_ab = {"us_per_step": exchange_ab, "plan_of_the_run": plan_now,
                       "steps_each": max(4, min(args.steps, 20)),
                       "note": "the run's own plan: its timed regions; the others: one region each after 3 "
                               "warm-up steps, maximum over ranks"}

    # ---- per-kernel durations: HIP events on the launch stream, same steps replayed -------
    # (two lanes whose steps are several kernels -- shapes outside the deferred statistics' range --
    # are replayed one launch at a time: per-kernel durations as rocprofv3 lists them for --lanes 1)
    fence()
    replay_one_lane = lanes[0] > 1 and not carried_flag
    if replay_one_lane:
        lanes[0] = 1
        _ffi.check(L.trlda_model_set_stream_lanes(model, 1))
        for i in range(4):
            step(pos[0] + i)
        pos[0] += 4
        one_lane_s = sorted(timed() for _ in range(3))[1]
    _ffi.check(L.trlda_model_set_timing(model, 1))
    for i in range(args.steps):
        step(pos[0] + i)
    pos[0] += args.steps
    fence()
    kernel_us = []
    for w in range(5):
        us, cnt = C.c_double(), C.c_int64()
        _ffi.check(L.trlda_model_get_timing(model, w, C.byref(us), C.byref(cnt)))
        kernel_us.append(us.value / max(cnt.value, 1))
    # Interval 4 holds no kernel (what a pair of event records costs on an idle stream
    # position; reported, not used).  The launches of a step run back to back, so what the
    # instrumented replay takes longer than the timed steps, spread over its launches, is the
    # events' share of every interval: taken out, so that the figures agree with rocprofv3's
    # kernel durations (profiles/*_kernel_stats.csv).
    event_pair_us = kernel_us.pop()
    carried = carried_flag
    if carried:
        # one launch per step: the other intervals hold no kernel (two event records back to back)
        kernel_us = [0.0, 0.0, kernel_us[2], 0.0]
    launches = [u for u in kernel_us if u > 0.5 * event_pair_us]
    step_us = 1e6 * (one_lane_s if replay_one_lane else elapsed) / args.steps   # (of the replayed form)
    event_us = 0.0
    laned = bool(carried and lanes[0] > 1 and L.trlda_model_lane_steps(model) > 0)
    if not collective and launches and not laned:
        event_us = min(max((sum(launches) - step_us) / len(launches), 0.0), event_pair_us)
    kernel_us = [max(u - event_us, 0.0) if u > 0.5 * event_pair_us else 0.0 for u in kernel_us]
    if laned:
        # two lanes: a lane's launches run back to back on its stream -- one pair of events per lane
        # around the replay's whole stretch, divided by the lane's launches (no event between two
        # launches: that would change how they overlap)
        us, cnt = C.c_double(), C.c_int64()
        _ffi.check(L.trlda_model_get_lane_timing(model, C.byref(us), C.byref(cnt)))
        kernel_us = [0.0, 0.0, us.value / max(cnt.value, 1), 0.0]
    _ffi.check(L.trlda_model_set_timing(model, 0))
    if replay_one_lane:
        lanes[0] = args.lanes
        _ffi.check(L.trlda_model_set_stream_lanes(model, lanes[0]))

    # executed iterations per document, every batch: mean, and the fp64 work of the document
    # kernel, sum_d I_d (4 K n_d + c_psi K) with c_psi = 91 fp64 operations per exp(psi) as
    # implemented (the two chains of the rational part in u = x (x + 9): 13 instructions, one reciprocal, the
    # seven-term series, exp; the chains in x of rounds 2-5a: 106;
    # a fused multiply-add counted as two: the five-reciprocal-pair form of round 1 was 140)
    # (csrc/psi.h; SURVEY.md 8d's secondary figure) plus the first phinorm (2 K n_d)
    C_PSI = 91.
    doc_flops, iter_sum = [], 0.
    for j in range(args.num_batches):
        step(j, want_iters=True)
        fence()
        it = iters_dev.cpu().numpy().astype(np.float64)
        n_d = np.diff(csrs[j].indptr).astype(np.float64)
        doc_flops.append(float((it * (4. * K * n_d + C_PSI * K) + 2. * K * n_d + C_PSI * K).sum()))
        iter_sum += float(it.mean())
    mean_iters = iter_sum / args.num_batches
    doc_flops = float(np.mean(doc_flops))

    word_sharded = bool(collective and exchange == "factors" and L.trlda_model_last_word_sharded(model))
    if rank != 0:
        dist.destroy_process_group()
        return

    estep_bytes, docs_bytes, _ = algorithmic_bytes(K, V, csrs[0].indptr)
    all_e, all_d = zip(*[algorithmic_bytes(K, V, c.indptr)[:2] for c in csrs])
    estep_bytes, docs_bytes = float(np.mean(all_e)), float(np.mean(all_d))
    docs_us = kernel_us[2]
    # with the next batch announced, the document-kernel launch also carries that batch's
    # preamble: 8 K V (row sums: lambda read once) + 16 K n_active (lambda in, exp(psi) out)
    pre_bytes = 0.0
    if prefetch and L.trlda_model_last_preamble_fused(model):
        pre_bytes = float(np.mean([8. * K * V + 16. * K * len(np.unique(c.ids)) for c in csrs]))
    docs_only_bytes = docs_bytes
    docs_bytes = docs_bytes + pre_bytes
    # ... and, with deferred statistics, the statistics of the step before: exp(psi(lambda)) of
    # its active words in, K x V statistics out, the rows of exp(psi(gamma)) gathered once per
    # entry + (document, weight) per entry (kernel 4) -- together with the
    # two shares above one whole E-step of SURVEY.md 8(d)'s bytes_alg per launch
    stats_bytes = 0.0
    if carried:
        stats_bytes = float(np.mean([8. * K * len(np.unique(c.ids)) + 8. * K * V +
                                     (8. * K + 12.) * float(c.indptr[-1]) for c in csrs]))
        docs_bytes = min(docs_bytes + stats_bytes, estep_bytes)
        stats_bytes = docs_bytes - docs_only_bytes - pre_bytes
    # Two lanes: two launches are in flight at any time, each waiting for the CUs the other one's
    # workgroups still hold -- a launch's duration (events around it on its lane's stream; what
    # rocprofv3 lists) is then about twice the time the device spends per launch.  The roofline is
    # priced on that time: the timed region divided by its launches (one per step).
    launch_us = docs_us
    if laned:
        docs_us = step_us
    achieved = docs_bytes / (docs_us * 1e-6) / 1e9 if docs_us > 0 else 0.0
    traffic, traffic_src = None, None
    # the document stage under the name rocprofv3 lists it by (profiles/*_kernel_stats.csv)
    doc_kernel = (L.trlda_model_last_doc_kernel(model) or b"estep_docs_kernel").decode()
    if carried:                                      # (estep_merged.h: the launch that also carries statistics)
        doc_kernel = doc_kernel.replace("_kernel", "_deferred_kernel")
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # measured for the default workload only
            names = tj.get("kernels") or [tj.get("kernel", "")]
            if any(n.endswith(doc_kernel) for n in names) and (K, V, args.batch) == (100, 7000, 200) \
                    and args.lengths == "poisson" and not args.uniform:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = {k: tj.get(k) for k in ("measured_at_commit", "fetch_size_kb",
                                                      "write_size_kb") if k in tj}
        except (OSError, ValueError):
            traffic = None
    kernel_names = list(KERNEL_NAMES)
    kernel_names[2] = doc_kernel
    if args.sstats_mode == "atomic":
        kernel_names[3] = "elementwise_stream_kernel<FinishOp>"
    elif K <= 512:
        kernel_names[3] = "sstats_update_kernel"
    kernel_pairs = list(zip(kernel_names, kernel_us))
    if L.trlda_model_last_preamble_fused(model):      # kernels 1 and 2 ran as one launch
        kernel_pairs = [("preamble_fused_kernel", kernel_us[0])] + kernel_pairs[2:]
        if prefetch:
            # ... or as workgroups [n_docs, ..) of the previous step's document-kernel launch
            kernel_pairs = [(doc_kernel + " (documents of this step + preamble workgroups for "
                             "the next step's batch)", kernel_us[2])] + kernel_pairs[2:]
        if carried:
            kernel_pairs = [(doc_kernel + " (documents of this step + preamble workgroups for the next "
                             "step's batch + statistics workgroups for the step before)", kernel_us[2])]
    roofline = {
        "bound": "hbm", "kernel": "trlda::" + doc_kernel,
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        # the same launch counted for its documents only (the next batch's preamble rides along
        # in the numerator of `frac`, not in the launch's duration: DESIGN.md 5)
        "frac_documents_only": round(docs_only_bytes / (docs_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
        if docs_us > 0 else 0.0,
        "traffic": traffic,
        # counters need rocprofv3 passes of their own (MI355X_MICROARCH.md): the figure is read from
        # profiles/traffic.json, stamped with the commit it was measured at -- not measured in this run
        "traffic_in_run": False,
        "traffic_source": traffic_src,
        "algorithmic_bytes_per_launch": docs_bytes,
        "algorithmic_bytes_split": {"documents": docs_only_bytes, "next_batch_preamble": pre_bytes,
                                    "previous_batch_statistics": stats_bytes},
        "avg_launch_us": round(launch_us, 2),
        "launches_in_flight": round(launch_us / docs_us, 2) if laned and docs_us > 0 else 1,
        "device_time_per_launch_us": round(docs_us, 2),
        "method": ("two stream lanes: `achieved` = algorithmic bytes per launch / (timed region / its "
                   "launches); avg_launch_us = HIP events on the lanes' streams in a "
                   "replay of the timed steps, one pair of events per lane around its whole stretch of "
                   "back-to-back launches -- launches of the two lanes overlap, so it is launches_in_flight "
                   "times the device time per launch (rocprofv3's per-launch duration is this figure; "
                   "`--lanes 1`: one launch at a time, value_one_lane)") if laned else
                  "HIP events on the launch stream around every launch in a replay of the timed "
                  "steps, minus the events' own share (replay time over timed time, per launch)",
        "event_share_us": round(event_us, 2), "empty_event_pair_us": round(event_pair_us, 2),
        "kernels_us": {n: round(u, 2) for n, u in kernel_pairs},
        # HBM is not what bounds this kernel (its traffic is below the algorithmic bytes): the same
        # launch against the fp64 vector peak (MI355X_MICROARCH.md: 157.3 TF fp32 vector, fp64 at
        # half of it)
        "fp64": {"flops_per_launch": doc_flops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "achieved": round(doc_flops / (docs_us * 1e-6) / 1e12, 3) if docs_us > 0 else 0.0,
                 "frac": round(doc_flops / (docs_us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS, 5)
                 if docs_us > 0 else 0.0,
                 "model": "sum_d I_d (4 K n_d + 91 K) + 2 K n_d + 91 K"},
        "estep": {   # the whole path against SURVEY.md 8(d)'s bytes_alg
            "algorithmic_bytes_per_step": estep_bytes,
            "achieved": round(estep_bytes * args.steps / elapsed / 1e9, 2),
            "frac": round(estep_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5)},
    }

    # ---- secondary: whole update_parameters calls through the Python surface (N = 1) --------
    update_rates = None
    if world == 1 and not args.no_update_rates:
        from trlda_amd.models import OnlineLDA
        om = OnlineLDA.__new__(OnlineLDA)
        om._num_documents, om._update_count = 1000000, 0
        om._ada_tau, om._ada_rho, om._ada_sq_norm = 1000., 1e-3, 1.
        om._setup(V, K, .1, .3, local_rank, _lambda=lam)
        # a mini-batch of its own for every call: a model fed one mini-batch again and again fits
        # it and its E-steps then leave through the convergence test after a few iterations
        # (rounds 1-2 timed that: 0.33 instead of 0.55 ms per call at the headline shape)
        n_dev, n_lst = (30, 10) if B <= 1600 else (6, 3)
        upd = [rank_corpus(0, 500 + i) for i in range(n_dev + 1)]
        resident = [om.upload(c) for c in upd]
        as_list = [c.to_list() for c in upd[:n_lst + 1]]
        update_rates = {"unit": "docs/s", "note": "update_parameters(docs, max_iter_inference=%d) "
                        "end to end from Python (gamma0 drawn per call, lda.cpp:135; the list form "
                        "includes flattening and upload; every call gets a mini-batch of its own, "
                        "every series starts from the same lambda); measured before the CPU "
                        "baseline's threads start; never `value`" % args.max_iter}
        for label, seq in (("device_batch", resident), ("list_of_tuples", as_list)):
            for tr in (0, 10):
                _ffi.check(L.trlda_model_set_lambda(om._handle, lam))
                om.update_parameters(seq[0], max_iter_tr=tr, max_iter_inference=args.max_iter)
                _ffi.check(L.trlda_model_synchronize(om._handle))
                t_u = time.perf_counter()
                for docs_in in seq[1:]:
                    om.update_parameters(docs_in, max_iter_tr=tr, max_iter_inference=args.max_iter)
                _ffi.check(L.trlda_model_synchronize(om._handle))
                dt_u = (time.perf_counter() - t_u) / (len(seq) - 1)
                update_rates["%s_tr%d" % (label, tr)] = {"docs_per_s": round(B / dt_u, 1),
                                                        "ms_per_call": round(1e3 * dt_u, 4)}
        for r in resident:
            r.close()
        om.close()

    # ---- parity + CPU baseline (rank 0, N = 1 only): the checker, timed beside the GPU ----
    cpu_baseline, parity = None, None
    if collective:
        _ffi.check(L.trlda_model_set_lambda(model, lam))      # the parity leg wants lambda0 back
    if world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle                   # test infrastructure: checker + baseline
        orc = pyoracle.Oracle()
        c = csrs[0]
        g0 = np.asfortranarray(gamma0s[0].cpu().numpy().T)
        go, so, ito = orc.estep(lam, .1, c.indptr, c.ids, c.cnts, g0, args.max_iter,
                                args.threshold, nthreads=min(8, os.cpu_count() or 1))
        step(0, want_iters=True, plain=True)
        fence()
        gg = np.asfortranarray(gamma.cpu().numpy().reshape(B, K).T)
        sg = sstats.cpu().numpy().reshape(K, V, order="F")
        nz = so > 0
        parity = {"gamma_max_rel_err": float(np.max(np.abs(gg - go) / np.abs(go))),
                  "sstats_max_rel_err": float(np.max(np.abs(sg[nz] - so[nz]) / so[nz])),
                  "iteration_counts_equal": bool(np.array_equal(iters_dev.cpu().numpy(), ito)),
                  "against": "oracle/cpu_ref.c on the first timed batch"}

        def time_cpu(fn, budget):
            n, t_start = 0, time.perf_counter()
            while True:
                cc = csrs[n % len(csrs)]
                g_init = np.asfortranarray(gamma0s[n % len(csrs)].cpu().numpy().T)
                fn(cc, g_init)
                n += 1
                dt = time.perf_counter() - t_start
                if dt >= budget or n >= 400:
                    return n, dt

        if not args.parity_only:
            ncores = os.cpu_count() or 1
            if pyoracle.Reference.available():
                ref = pyoracle.Reference()
                rm = ref.online(V, K, 1000000, alpha=.1, eta=.3)
                rm.lambdas = lam
                n1, t1 = time_cpu(lambda cc, g: rm.estep(cc.indptr, cc.ids, cc.cnts, g, args.max_iter,
                                                         args.threshold), args.cpu_seconds)
                kind = "reference"
            else:
                n1, t1 = time_cpu(lambda cc, g: orc.estep(lam, .1, cc.indptr, cc.ids, cc.cnts, g,
                                                          args.max_iter, args.threshold),
                                  args.cpu_seconds)
                kind = "port"
            # doc-parallel port: a 200-document batch does not feed hundreds of threads; take the
            # best of a few thread counts (each a short sample)
            best = None
            for nthr in sorted({t for t in (8, 16, 32, 64, ncores) if t <= ncores}):
                nm, tm = time_cpu(lambda cc, g: orc.estep(lam, .1, cc.indptr, cc.ids, cc.cnts, g,
                                                          args.max_iter, args.threshold,
                                                          nthreads=nthr), args.cpu_seconds / 8)
                if best is None or nm / tm > best[0] / best[1]:
                    best = (nm, tm, nthr)
            nm, tm, best_threads = best
            cpu_baseline = {
                "value": round(n1 * B / t1, 1), "unit": "docs/s", "cores": 1, "kind": kind,
                "sample": "%d E-step calls over the bench's own %d-document mini-batches "
                          "(%.1f s), single thread" % (n1, B, t1),
                "all_cores": {"value": round(nm * B / tm, 1), "cores": best_threads,
                              "cores_available": ncores, "kind": "port",
                              "note": "oracle/cpu_ref.c doc-parallel variant, best of 8..%d threads; "
                                      "the reference's own OpenMP path is slower than its single "
                                      "thread (BASELINE.md)" % ncores},
                "gpu_over_cpu_1thread": round(docs_per_s / (n1 * B / t1), 1),
            }
            try:
                cpu_baseline["cpu_model"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo")
                                             if l.startswith("model name")][0]
            except Exception:
                pass

    out = {
        "metric": "E-step docs/sec (mini-batch) at K=100, V=7000",
        "value": round(docs_per_s, 1), "unit": "docs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 5),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "OnlineLDA E-step (LDA::updateVariablesVI) K=%d V=%d batch=%d "
                               "docs/GPU, max_iter_inference=%d, threshold=%g, %s vocabulary, "
                               "~%d unique words/doc%s" % (K, V, B, args.max_iter, args.threshold,
                                                           "uniform" if args.uniform else
                                                           "Zipf-1.07", args.mean_unique,
                                                           " (log-normal lengths, longest %d)"
                                                           % max(int(np.diff(c.indptr).max()) for c in csrs)
                                                           if args.lengths == "lognormal" else ""),
                   "num_topics": K, "num_words": V, "batch_per_gpu": B, "global_batch": B * world,
                   "num_batches": args.num_batches,
                   "documents_streamed": args.num_batches * B * world,
                   "max_iter_inference": args.max_iter, "threshold": args.threshold,
                   "mean_iterations_executed": round(mean_iters, 2),
                   "sstats": args.sstats_mode,
                   "exp_elog_beta": "all V words" if args.dense_preamble else
                   "active words of the batch (mean %d of %d)" % (
                       int(np.mean([len(np.unique(c.ids)) for c in csrs])), V),
                   "preamble": ("prepared by extra workgroups of the previous step's document-kernel "
                                "launch (trlda_model_estep_io_next)" if prefetch else
                                "a kernel launch of its own every step"),
                   "statistics": ("formed by extra workgroups of the NEXT step's document-kernel launch "
                                  "(trlda_model_set_deferred_stats); the last step's by a kernel of their "
                                  "own, inside the timed region" if carried else
                                  "a kernel launch of its own every step"),
                   "in_flight": ("two E-steps at a time: the steps go in turn to two streams of the library's "
                                 "own (trlda_model_set_stream_lanes(2), trlda_model_estep_io_ahead); output "
                                 "arrays alternate between two sets; joined by every fence"
                                 if (laned or replay_one_lane) else "one E-step at a time"),
                   "parallelism": "dp%d" % world,
                   "exchange_via": (("trlda_model_estep_dp (direct: peers' buffers through hipIpc, a step "
                                     "counter per source)" if exchange == "factors" and direct else
                                     "trlda_model_estep_dp (ncclAllGather on the model's stream)"
                                     if exchange == "factors" else
                                     "trlda_model_allreduce_sstats (own ncclComm_t)" if rccl_comm is not None
                                     else "torch.distributed.all_reduce") if collective else None),
                   "exchange": ((("all-gather of the documents' factors (expElogtheta rows + entry weights, "
                                  "%.2f MB per rank and step); every rank forms statistics + M-step "
                                  "(onlinelda.cpp:99-100) for ITS range of the vocabulary and the ranks exchange "
                                  "the lambda columns they wrote, in place (%.2f MB received per rank and step)"
                                  % (8e-6 * xworld * slot, 8e-6 * KV * (xworld - 1) / xworld))
                                 if word_sharded else
                                 ("all-gather of the documents' factors (expElogtheta rows + entry weights, "
                                  "%.2f MB per rank and step); every rank forms the statistics of the whole "
                                  "%d-document mini-batch and the M-step (onlinelda.cpp:99-100) in one kernel"
                                  % (8e-6 * xworld * slot, B * xworld))) if exchange == "factors" else
                                "RCCL all-reduce of K x V fp64 sstats, then the M-step "
                                "(%s) that the next step's E-step reads"
                                % ("batchlda.cpp:60" if args.batch_lda else "onlinelda.cpp:99-100"))
                   if collective else "none",
                   "exchange_check": exchange_probe,
                   "word_sharded_m_step": word_sharded if collective else None,
                   "virtual_world": vworld or None},
        "repeats": repeats,
        # the untimed, declared settle phase in front of the first timed region (settle(), above)
        "settle_steps": settled["settle_steps"], "settle_ms": settled["settle_ms"],
        "settle": dict(settled, rule="samples of min(steps, 20) steps until two consecutive ones agree "
                                     "within 2 %% and %g s have passed, at most %g s; untimed; --no-settle "
                                     "switches it off" % (SETTLE_MIN_S, SETTLE_MAX_S)),
        # what kind of step `value` times, so that lines of different rounds / switches are compared like
        # for like: rounds 1-4 = {"deferred_stats": false, "lanes": 1}; the synchronous
        # forms are `value_one_lane` (absent under --headline-only, whose runs are profiled per kernel)
        # and `--no-deferred --lanes 1`
        "mode": {"deferred_stats": bool(carried), "lanes": lanes[0], "announced_preamble": bool(prefetch),
                 "pipelined": bool(carried or laned),
                 "results_complete": "at the fence that closes the timed region (trlda_model_flush + "
                                     "synchronisation), inside the clock" if (carried or laned) else
                                     "when each step's kernels end"},
        "value_no_prefetch": value_no_prefetch,
        "value_one_lane": value_one_lane,
        "lanes": lanes[0],
        # what became of the lanes (include/trlda_hip.h, trlda_model_lane_state: 2 = two lanes on streams
        # seen to run side by side, 1 = given up) and what the library's own measurement found
        "lane_state": lane_state, "lane_calibration": lane_cal,
        "value_fixed_work": value_fixed_work,
        "value_end_to_end": value_end_to_end,
        "rccl_ranks": rccl_ranks,
        "exchange_ab": exchange_ab,
        "same_step_n1": same_step_n1,
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "parity": parity,
        "update_parameters": update_rates,
    }
    if collective and not vworld:
        dist.destroy_process_group()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)                         # (C stdio of the libraries: to stderr, above)
    os.write(result_fd, (json.dumps(out) + "\n").encode())   # the one JSON line on stdout
    os.close(result_fd)


if __name__ == "__main__":
    main()
