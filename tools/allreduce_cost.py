"""World-size-1 cost of the path's exchange step on this GPU: an RCCL all-reduce of the K x V
fp64 statistics at BASELINE.json's three table sizes (5.6 MB, 80 MB, 400 MB), through
torch.distributed and through the C ABI (trlda_model_allreduce_sstats with a communicator from
ncclCommInitRank).  With one rank nothing crosses a link: this is the latency floor (launch +
RCCL's own kernel) that every N > 1 step pays on top of the xGMI transfer.

    python tools/allreduce_cost.py          (GPU box, repo root)
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")

import torch
import torch.distributed as dist

from trlda_amd import _ffi

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
L = _ffi.lib()
rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
uid, comm = UniqueId(), C.c_void_p()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0

for K, V in ((100, 7000), (200, 50000), (500, 100000)):
    n = K * V
    buf = torch.ones(n, dtype=torch.float64, device=dev)
    model = _ffi.vp()
    _ffi.check(L.trlda_model_create(C.byref(model), 0, K, V))
    stream = torch.cuda.current_stream(dev).cuda_stream
    _ffi.check(L.trlda_model_set_stream(model, _ffi.vp(stream)))
    reps = 200 if n < 10 ** 7 else 30
    for label, fn in (("torch.distributed.all_reduce", lambda: dist.all_reduce(buf)),
                      ("trlda_model_allreduce_sstats", lambda: _ffi.check(
                          L.trlda_model_allreduce_sstats(model, comm, C.c_void_p(buf.data_ptr()))))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / reps
        print("K=%d V=%d  %7.1f MB  %-30s  %8.1f us/call on the stream, %8.1f us wall"
              % (K, V, n * 8 / 1e6, label, e0.elapsed_time(e1) * 1e3 / reps, wall * 1e6))
    L.trlda_model_destroy(model)
dist.destroy_process_group()
