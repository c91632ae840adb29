"""A RCCL communicator of the process's own, for the C ABI's collectives.

``trlda_model_allreduce_sstats`` / ``trlda_model_online_update_dp`` take the host program's
``ncclComm_t`` and enqueue the collective on the model's stream themselves; torch keeps its
communicators private, so the Python host makes one: rank 0's ``ncclGetUniqueId`` travels through
the (already initialised) ``torch.distributed`` group, every rank calls ``ncclCommInitRank``.
The library loaded is the RCCL torch ships, the one already in the process.
"""
import ctypes as C
import os


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_LIB = None


def _librccl():
    global _LIB
    if _LIB is None:                                  # one handle per process
        import torch
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = C.CDLL(path if os.path.exists(path) else "librccl.so", mode=C.RTLD_GLOBAL)
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _LIB = lib
    return _LIB


def own_communicator(dist, device, group=None):
    """``ncclComm_t`` (a ``ctypes.c_void_p``) over the ranks of ``group``; raises on failure.
    Collective: every rank of the group must call it."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lib = _librccl()
    uid = _UniqueId()
    # (a failure on rank 0 travels with the id: no rank is left waiting in the broadcast)
    good = 1
    if rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0:
        good = 0
    raw = torch.frombuffer(bytearray(bytes(uid) + bytes([good])), dtype=torch.uint8).clone().to(device)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(raw, src=src, group=group)
    host = raw.cpu().numpy().tobytes()
    if host[128] != 1:
        raise RuntimeError("ncclGetUniqueId failed on rank 0")
    C.memmove(C.byref(uid), host[:128], 128)
    comm = C.c_void_p()
    if lib.ncclCommInitRank(C.byref(comm), world, uid, rank) != 0:
        raise RuntimeError("ncclCommInitRank failed")
    return comm


def comm_count(comm):
    """Number of ranks RCCL itself reports for the communicator (ncclCommCount); -1 on failure."""
    n = C.c_int(-1)
    if not comm or _librccl().ncclCommCount(comm, C.byref(n)) != 0:
        return -1
    return n.value


def destroy(comm):
    if comm:
        _librccl().ncclCommDestroy(comm)
