"""One rank of a data-parallel run with FACTOR exchange (csrc/dp_kernels.h), as its own process on
the one GPU of the test box: test infrastructure for tests/test_gpu_dp.py.

RCCL refuses two ranks on one device, so the ranks exchange their slots through the hook of
trlda_model_set_allgather ("a host with a transport of its own"): device -> a shared-memory
file -> every rank's device buffer, with a spin barrier on per-rank arrival counters.  Everything
else -- one process, one libc generator, one model per rank -- is what a real N-GPU run has.

usage: dp_worker.py <config.json> <rank>
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class FileTransport(object):
    """all-gather among the processes that map the same file"""

    def __init__(self, path, rank, world, max_count, lib, device):
        self.rank, self.world, self.lib, self.device = rank, world, lib, device
        self.path = path
        self.arrive = np.memmap(path + ".bar", dtype=np.int64, mode="r+", shape=(world,))
        self.slots = np.memmap(path + ".dat", dtype=np.float64, mode="r+", shape=(world, max_count))
        self.epoch = 0
        self.calls = 0
        self.bytes = 0
        self.table = None               # the shared K x V table of the word-sharded M-step
        self.vcalls = 0
        self.vbytes = 0

    def gatherv(self, ctx, table, offsets, rank, world, stream):
        """trlda_allgatherv_fn: every rank wrote its range [offsets[r], offsets[r + 1]) of `table`
        (doubles) and receives the others', in place"""
        try:
            L = self.lib
            offs = [int(offsets[r]) for r in range(world + 1)]
            assert rank == self.rank and world == self.world
            if self.table is None:
                self.table = np.memmap(self.path + ".lam", dtype=np.float64, mode="r+", shape=(offs[-1],))
            if L.trlda_dev_synchronize(self.device) != 0:      # this rank's columns are complete
                return 1
            lo, hi = offs[rank], offs[rank + 1]
            if hi > lo:
                mine = self.table[lo:hi]
                if L.trlda_dev_download(self.device, C.c_void_p(mine.ctypes.data),
                                        C.c_void_p(table + lo * 8), (hi - lo) * 8) != 0:
                    return 2
                self.table.flush()
            self.barrier()
            for r in range(world):
                lo, hi = offs[r], offs[r + 1]
                if r == rank or hi == lo:
                    continue
                src = np.ascontiguousarray(self.table[lo:hi])
                if L.trlda_dev_upload(self.device, C.c_void_p(table + lo * 8), C.c_void_p(src.ctypes.data),
                                      (hi - lo) * 8) != 0:
                    return 3
            self.barrier()                 # nobody rewrites its range before everyone has read it
            self.vcalls += 1
            self.vbytes += (offs[-1] - (offs[rank + 1] - offs[rank])) * 8
            return 0
        except Exception as exc:           # noqa: BLE001 -- a raise cannot cross the C frame
            sys.stderr.write("gatherv transport failed: %r\n" % (exc,))
            return 9

    def barrier(self):
        self.epoch += 1
        self.arrive[self.rank] = self.epoch
        self.arrive.flush()
        deadline = time.time() + 300
        while int(np.min(np.asarray(self.arrive))) < self.epoch:
            if time.time() > deadline:
                raise RuntimeError("barrier timed out")
            time.sleep(0.0002)

    def __call__(self, ctx, send, recv, count, stream):
        try:
            L = self.lib
            count = int(count)
            assert count <= self.slots.shape[1], (count, self.slots.shape)
            assert send == recv + self.rank * count * 8        # the in-place form
            if L.trlda_dev_synchronize(self.device) != 0:      # this rank's slot is complete
                return 1
            mine = self.slots[self.rank, :count]
            if L.trlda_dev_download(self.device, C.c_void_p(mine.ctypes.data), C.c_void_p(send),
                                    count * 8) != 0:
                return 2
            self.slots.flush()
            self.barrier()
            for r in range(self.world):
                if r == self.rank:
                    continue
                src = np.ascontiguousarray(self.slots[r, :count])
                if L.trlda_dev_upload(self.device, C.c_void_p(recv + r * count * 8),
                                      C.c_void_p(src.ctypes.data), count * 8) != 0:
                    return 3
            self.barrier()                 # nobody refills its slot before everyone has read it
            self.calls += 1
            self.bytes += (self.world - 1) * count * 8
            return 0
        except Exception as exc:           # noqa: BLE001 -- a raise cannot cross the C frame
            sys.stderr.write("transport failed: %r\n" % (exc,))
            return 9


def main():
    cfg = json.load(open(sys.argv[1]))
    rank = int(sys.argv[2])
    world = cfg["world"]
    K, V, D = cfg["K"], cfg["V"], cfg["D"]

    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments, DeviceBatch
    from trlda_amd.utils.synthetic import make_corpus
    L = _ffi.lib()
    _ffi.require_gpu()
    dev = 0

    model = _ffi.vp()
    _ffi.check(L.trlda_model_create(C.byref(model), dev, K, V))
    rng = np.random.RandomState(cfg["lambda_seed"])
    lam = np.asfortranarray(rng.gamma(100., .01, (K, V)))
    _ffi.check(L.trlda_model_set_lambda(model, lam))
    _ffi.check(L.trlda_model_set_alpha(model, np.full(K, cfg["alpha"])))
    if cfg.get("plain"):
        _ffi.check(L.trlda_model_set_fused_update(model, 0))
        _ffi.check(L.trlda_model_set_carry_rowsums(model, 0))

    transport = FileTransport(cfg["path"], rank, world, cfg["max_count"], L, dev)
    HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    hook = HOOK(transport)
    if cfg.get("direct"):
        # the direct exchange: every rank exports its region (hipIpc), the 64-byte handles travel
        # through files, every rank maps its peers' regions; no hook, no collective afterwards
        mine = C.create_string_buffer(64)
        _ffi.check(L.trlda_model_dp_direct_alloc(model, cfg["max_count"], world, mine))
        with open(cfg["path"] + ".handle%d" % rank, "wb") as f:
            f.write(mine.raw)
        transport.barrier()
        handles = b"".join(open(cfg["path"] + ".handle%d" % r, "rb").read() for r in range(world))
        assert len(handles) == 64 * world
        _ffi.check(L.trlda_model_dp_direct_connect(model, rank, world, handles))
        transport.barrier()
    else:
        _ffi.check(L.trlda_model_set_allgather(model, C.cast(hook, C.c_void_p), None))
    # the word-sharded M-step's transport (ranges of unequal size, in place) -- unless the run is
    # to form the whole mini-batch's statistics on every rank
    HOOKV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_void_p)
    hookv = HOOKV(transport.gatherv)
    if cfg.get("word_sharded", True) and not cfg.get("direct"):
        _ffi.check(L.trlda_model_set_allgatherv(model, C.cast(hookv, C.c_void_p), None))

    out = {}
    count = C.c_int(0)
    for call, spec in enumerate(cfg["calls"]):
        csr = CSRDocuments(*make_corpus(spec["B"], V, seed=spec["corpus_seed"],
                                        mean_unique=spec.get("mean_unique", 60),
                                        lengths=spec.get("lengths")))
        cuts = np.asarray(spec["cuts"], dtype=np.int32) if "cuts" in spec \
            else csr.shard_cuts(world).astype(np.int32)
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        batch = DeviceBatch(csr, V, dev)
        shard = DeviceBatch(csr.slice(lo, hi), V, dev)
        cuts_p = cuts.ctypes.data_as(C.POINTER(C.c_int32))
        if spec["kind"] == "update":
            L.trlda_seed(spec["seed"])
            rho = C.c_double(0.)
            _ffi.check(L.trlda_model_online_update_dp(
                model, batch.handle, shard.handle, None, rank, world, cuts_p, D, cfg["eta"],
                spec["max_iter_tr"], spec["max_iter_inference"], .7, 100., -1., 1, 1e-3,
                C.byref(count), C.byref(rho)))
            out["rho%d" % call] = np.array([rho.value])
        elif spec["kind"] == "batch":                       # BatchLDA epochs, batchlda.cpp:43-61
            L.trlda_seed(spec["seed"])
            _ffi.check(L.trlda_model_batch_update_dp(
                model, batch.handle, shard.handle, None, rank, world, cuts_p, cfg["eta"],
                spec["max_epochs"], spec["max_iter_inference"], 1, 1e-3))
        else:                                               # one E-step from a given gamma0
            g0 = np.load(spec["gamma0"])                    # K x B, column-major file
            mine = np.ascontiguousarray(g0[:, lo:hi].T)     # docs_r x K == K x docs_r col-major
            nb = max(hi - lo, 1) * K * 8
            g_dev, s_dev, it_dev = _ffi.vp(), _ffi.vp(), _ffi.vp()
            _ffi.check(L.trlda_dev_alloc(dev, nb, C.byref(g_dev)))
            _ffi.check(L.trlda_dev_alloc(dev, K * V * 8, C.byref(s_dev)))
            _ffi.check(L.trlda_dev_alloc(dev, max(hi - lo, 1) * 4, C.byref(it_dev)))
            if hi > lo:
                _ffi.check(L.trlda_dev_upload(dev, g_dev, C.c_void_p(mine.ctypes.data), mine.nbytes))
            mstep = int(spec.get("mstep", 0))
            _ffi.check(L.trlda_model_estep_dp(
                model, batch.handle, shard.handle, None, rank, world, cuts_p, None, g_dev, s_dev,
                spec["max_iter"], 1e-3, it_dev, mstep, None, spec.get("rho", 0.), cfg["eta"],
                spec.get("scale", 1.)))
            _ffi.check(L.trlda_model_synchronize(model))
            s = np.empty((K, V), order="F")
            _ffi.check(L.trlda_dev_download(dev, C.c_void_p(s.ctypes.data), s_dev, s.nbytes))
            g = np.empty((hi - lo, K))
            its = np.empty(hi - lo, dtype=np.int32)
            if hi > lo:
                _ffi.check(L.trlda_dev_download(dev, C.c_void_p(g.ctypes.data), g_dev, g.nbytes))
                _ffi.check(L.trlda_dev_download(dev, C.c_void_p(its.ctypes.data), it_dev, its.nbytes))
            out["sstats%d" % call] = s
            out["gamma%d" % call] = g.T
            out["iters%d" % call] = its
            for p in (g_dev, s_dev, it_dev):
                L.trlda_dev_free(dev, p)
        batch.close()
        shard.close()
    lam_out = np.empty((K, V), order="F")
    _ffi.check(L.trlda_model_get_lambda(model, lam_out))
    out["lambda"] = lam_out
    out["exchanges"] = np.array([transport.calls, transport.bytes])
    out["lambda_exchanges"] = np.array([transport.vcalls, transport.vbytes])
    out["word_sharded"] = np.array([L.trlda_model_last_word_sharded(model)])
    out["update_count"] = np.array([count.value])
    np.savez(cfg["path"] + ".rank%d.npz" % rank, **out)
    _ffi.check(L.trlda_model_synchronize(model))
    transport.barrier()                    # nobody unmaps a region a peer may still write to
    L.trlda_model_destroy(model)
    print("DP-RANK-OK", rank)


if __name__ == "__main__":
    main()
