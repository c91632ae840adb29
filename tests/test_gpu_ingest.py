"""trlda_batch_create since round 6: the index (csrc/batch_index.cpp, held to golden digests by
tests/test_batch_index.py) is built and uploaded on worker threads while the caller goes on; the
first user of a batch takes over a build nobody has started; a batch destroyed unused is never
indexed.  Here, on the device: what the kernels see IS that index, bitwise, whoever built it and
in whatever order batches are made, used and dropped; E-steps on batches used the moment they are
created equal E-steps on batches that had time to be built; errors stay synchronous.
(reference: the Documents argument of LDA::updateVariables, python/src/ldainterface.cpp:152-190)"""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

from helpers import HipSampler, relerr, seeded_gamma, seeded_lambda

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_index_golden",
                                              os.path.join(ROOT, "tests", "golden", "make_index_golden.py"))
gold = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gold)


@pytest.fixture(scope="module")
def hip(hip_lib):
    from trlda_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a visible MI355X"
    return hip_lib


@pytest.fixture(scope="module")
def raw(hip):
    from trlda_amd import _ffi
    lib = C.CDLL(_ffi.LIB_PATH)
    lib.trlda_debug_batch_blob.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    lib.trlda_batch_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
    lib.trlda_batch_destroy.argtypes = [C.c_void_p]
    return lib


def host_index(raw, V, ip, ii, cc, cus):
    info = np.zeros(64, np.int64)
    p32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    pi = info.ctypes.data_as(C.POINTER(C.c_int64))
    f = raw.trlda_debug_batch_index
    f.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int,
                  C.POINTER(C.c_int64), C.c_void_p, C.c_size_t]
    assert f(V, len(ip) - 1, p32(ip), p32(ii), p32(cc), cus, pi, None, 0) == 0
    buf = np.zeros(int(info[24]), np.uint8)
    assert f(V, len(ip) - 1, p32(ip), p32(ii), p32(cc), cus, pi, buf.ctypes.data, buf.size) == 0
    head = {k: int(info[i]) for i, k in enumerate(gold.HEAD)}
    return buf, [int(v) for v in info[32:52]], gold.section_bytes(head)


def test_the_device_sees_the_host_index_whoever_built_it(hip, raw):
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    cases = {k: v for k, v in gold.cases().items() if k != "very_long_lists"}
    names = sorted(cases)
    handles, states = {}, {}
    # made back to back (the workers get behind), read in another order, some of them at once
    for n in names:
        V, (ip, ii, cc) = cases[n]
        ip, ii, cc = (np.array(a, dtype=np.int32, copy=True) for a in (ip, ii, cc))
        h = C.c_void_p()
        assert raw.trlda_batch_create(C.byref(h), 0, V, len(ip) - 1, ip.ctypes.data, ii.ctypes.data,
                                      cc.ctypes.data) == 0, n
        handles[n] = h
        ii[:] = -5                                   # (the caller's arrays are the caller's again)
        del ip, ii, cc
    for n in names[::-1]:
        V, (ip, ii, cc) = cases[n]
        want, offs, sizes = host_index(raw, V, np.ascontiguousarray(ip, np.int32), np.ascontiguousarray(ii, np.int32),
                                       np.ascontiguousarray(cc, np.int32), cus)
        got = np.empty_like(want)
        st = C.c_int(-1)
        assert raw.trlda_debug_batch_blob(handles[n], got.ctypes.data, got.size, C.byref(st)) == 0, n
        states[n] = st.value
        for name, o, nb in zip(gold.SECTIONS, offs, sizes):
            assert np.array_equal(got[o:o + nb], want[o:o + nb]), (n, name)
    for h in handles.values():
        raw.trlda_batch_destroy(h)
    assert set(states.values()) <= {0, 1, 3, 5, 6}   # built | queued | being indexed | indexed, not uploaded | being uploaded


def test_batches_dropped_before_and_while_they_are_built(hip, raw):
    """destroyed at once (never indexed), destroyed while a worker is on it, thousands in a row: no
    staging buffer is lost (the pool has two dozen) and the next batch is as good as any"""
    from trlda_amd.utils.synthetic import make_corpus
    V = 3000
    ip, ii, cc = make_corpus(150, V, seed=5, mean_unique=60)
    for r in range(3000):
        h = C.c_void_p()
        assert raw.trlda_batch_create(C.byref(h), 0, V, 150, ip.ctypes.data, ii.ctypes.data, cc.ctypes.data) == 0
        if r % 3 == 1:
            raw.trlda_batch_long_word_len(h)         # (takes the build over, or waits for the worker)
        if r % 7 == 3:
            import time
            time.sleep(2e-5)                         # (a worker has started on it)
        raw.trlda_batch_destroy(h)
    import torch
    want, offs, sizes = host_index(raw, V, ip, ii, cc, torch.cuda.get_device_properties(0).multi_processor_count)
    h = C.c_void_p()
    assert raw.trlda_batch_create(C.byref(h), 0, V, 150, ip.ctypes.data, ii.ctypes.data, cc.ctypes.data) == 0
    got = np.empty_like(want)
    assert raw.trlda_debug_batch_blob(h, got.ctypes.data, got.size, None) == 0
    for name, o, nb in zip(gold.SECTIONS, offs, sizes):
        assert np.array_equal(got[o:o + nb], want[o:o + nb]), name
    raw.trlda_batch_destroy(h)


@pytest.mark.timeout(300)
def test_a_corpus_of_batches_made_before_the_first_is_used(hip, raw):
    """The workers only index; the uploads are enqueued by the caller's thread -- trlda_batch_create for
    the batches made before it, an E-step for the batches announced to it, a batch's first user.  A
    caller that makes far more batches than there are staging buffers before it touches one of them
    (every mini-batch of an epoch up front) neither blocks nor loses one: every index reaches the device
    as the host builds it (ldainterface.cpp:152-190 -- the reference converts every document up front)."""
    import torch
    from trlda_amd.utils.synthetic import make_corpus
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    V, N = 2500, 150
    made = []
    for n in range(N):
        ip, ii, cc = make_corpus(60 + n % 7, V, seed=100 + n, mean_unique=40 + n % 5)
        h = C.c_void_p()
        assert raw.trlda_batch_create(C.byref(h), 0, V, len(ip) - 1, ip.ctypes.data, ii.ctypes.data, cc.ctypes.data) == 0
        made.append((h, ip, ii, cc))
    for n in list(range(0, N, 13)) + [N - 1, N - 2]:
        h, ip, ii, cc = made[n]
        want, offs, sizes = host_index(raw, V, ip, ii, cc, cus)
        got = np.empty_like(want)
        assert raw.trlda_debug_batch_blob(h, got.ctypes.data, got.size, None) == 0, n
        for name, o, nb in zip(gold.SECTIONS, offs, sizes):
            assert np.array_equal(got[o:o + nb], want[o:o + nb]), (n, name)
    for h, *_ in made:
        raw.trlda_batch_destroy(h)


@pytest.mark.timeout(300)
def test_four_threads_make_use_and_drop_batches_at_once(hip, raw):
    """The build tickets under real concurrency: four caller threads (ctypes releases the GIL) make
    batches ahead, read some at once and some late, drop some unused and some while a worker or another
    thread's trlda_batch_create is at them (that call uploads whatever index is finished, whoever made
    the batch).  Every index read back is the host's, bitwise; nothing hangs, nothing is lost
    (ldainterface.cpp:152-190)."""
    import threading
    import torch
    from trlda_amd.utils.synthetic import make_corpus
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    V = 2000
    cases = [make_corpus(40 + 11 * q, V, seed=900 + q, mean_unique=30 + 7 * q) for q in range(6)]
    want = [host_index(raw, V, *c, cus) for c in cases]
    errors = []

    def worker(t):
        try:
            rng = np.random.RandomState(t)
            window = []
            for n in range(400):
                q = int(rng.randint(len(cases)))
                ip, ii, cc = cases[q]
                h = C.c_void_p()
                assert raw.trlda_batch_create(C.byref(h), 0, V, len(ip) - 1, ip.ctypes.data, ii.ctypes.data,
                                              cc.ctypes.data) == 0
                window.append((h, q))
                act = rng.randint(4)
                if act == 0:                         # read the newest at once (takes the build over, or waits)
                    hh, qq = window.pop()
                elif act == 1 and len(window) > 6:   # read one made a while ago
                    hh, qq = window.pop(0)
                elif act == 2:                       # drop the newest unread
                    raw.trlda_batch_destroy(window.pop()[0])
                    continue
                else:
                    if len(window) > 10:
                        raw.trlda_batch_destroy(window.pop(0)[0])
                    continue
                buf, offs, sizes = want[qq]
                got = np.empty_like(buf)
                assert raw.trlda_debug_batch_blob(hh, got.ctypes.data, got.size, None) == 0
                for name, o, nb in zip(gold.SECTIONS, offs, sizes):
                    assert np.array_equal(got[o:o + nb], buf[o:o + nb]), (t, n, name)
                raw.trlda_batch_destroy(hh)
            for hh, _ in window:
                raw.trlda_batch_destroy(hh)
        except BaseException as exc:                 # noqa: BLE001 (reported by the main thread)
            errors.append((t, repr(exc)[:300]))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(240)
    assert not any(th.is_alive() for th in threads), "a thread hangs"
    assert not errors, errors


def test_errors_of_the_arguments_stay_with_the_call(hip, raw):
    ip = np.array([0, 2, 3], np.int32); cc = np.ones(3, np.int32)
    for bad in ([0, 9, 1], [0, -1, 1]):
        h = C.c_void_p()
        ii = np.array(bad, np.int32)
        assert raw.trlda_batch_create(C.byref(h), 0, 9, 2, ip.ctypes.data, ii.ctypes.data, cc.ctypes.data) == -3
        assert not h.value
    h = C.c_void_p()
    assert raw.trlda_batch_create(C.byref(h), 0, 9, 2, np.array([0, 2, 1], np.int32).ctypes.data,
                                  np.zeros(3, np.int32).ctypes.data, cc.ctypes.data) != 0


def test_used_at_once_or_after_a_while_the_same_e_step(hip, oracle):
    """an E-step on a batch the moment it is made (the call takes the build over) and on one that was
    made first, among thirty others: bitwise the same gamma and statistics, equal to the oracle's"""
    import time
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.models import OnlineLDA
    from trlda_amd.utils.synthetic import make_corpus
    sampler = HipSampler(hip)
    K, V, B = 50, 2000, 80
    lam = seeded_lambda(sampler, 4, K, V)
    g0 = seeded_gamma(sampler, 5, K, B)
    m = OnlineLDA.__new__(OnlineLDA)
    m._num_documents, m._update_count = 1000, 0
    m._ada_tau, m._ada_rho, m._ada_sq_norm = 1000., 1e-3, 1.
    m._setup(V, K, .1, .3, None, _lambda=lam)
    corpora = [CSRDocuments(*make_corpus(B, V, seed=900 + i, mean_unique=40)) for i in range(30)]
    early = [m.upload(c) for c in corpora]
    time.sleep(.01)
    a = [m.update_variables(b, latents=g0, max_iter=15) for b in early]
    b_ = [m.update_variables(m.upload(c), latents=g0, max_iter=15) for c in corpora]
    for (ga, sa), (gb, sb) in zip(a, b_):
        assert np.array_equal(ga, gb) and np.array_equal(sa, sb)
    c = corpora[7]
    go, so, _ = oracle.estep(lam, .1, c.indptr, c.ids, c.cnts, g0, 15, 1e-3)
    assert relerr(a[7][0], go) < 1e-9
    nz = so > 0
    assert relerr(a[7][1][nz], so[nz]) < 1e-9
    for b in early:
        b.close()
    m.close()


def test_a_corpus_pass_in_one_call(hip, oracle):
    """trlda_model_estep_corpus: a CSR corpus in host memory, the loop over its mini-batches inside the
    library (batches made eight ahead, indexed on the worker threads, deferred statistics, two lanes) -- gamma,
    iteration counts and every mini-batch's statistics bitwise those of a Python loop of do_e_step
    (python/src/ldainterface.cpp:311-390), a ragged last batch, a ring of statistics arrays shorter
    than the corpus; and against the oracle."""
    import torch
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.models import OnlineLDA
    from trlda_amd.stream import corpus_pass
    from trlda_amd.utils.synthetic import make_corpus
    sampler = HipSampler(hip)
    K, V, N, BS = 100, 3000, 1130, 200
    lam = seeded_lambda(sampler, 14, K, V)
    m = OnlineLDA.__new__(OnlineLDA)
    m._num_documents, m._update_count = 1000, 0
    m._ada_tau, m._ada_rho, m._ada_sq_norm = 1000., 1e-3, 1.
    m._setup(V, K, .1, .3, None, _lambda=lam)
    ip, ii, cc = make_corpus(N, V, seed=77, mean_unique=70)
    g0 = seeded_gamma(sampler, 15, K, N)
    dev = torch.device("cuda", 0)
    g0_d = torch.from_numpy(np.ascontiguousarray(g0.T)).to(dev)
    nb = (N + BS - 1) // BS
    for n_ring in (nb, 3):
        gam = torch.full((N, K), float("nan"), dtype=torch.float64, device=dev)
        its = torch.full((N,), -1, dtype=torch.int32, device=dev)
        ring = [torch.full((V, K), float("nan"), dtype=torch.float64, device=dev) for _ in range(n_ring)]
        corpus_pass(m, ip.astype(np.int64), ii, cc, BS, g0_d, gam, ring, max_iter=20, iterations=its)
        from trlda_amd import _ffi
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        got_g = gam.cpu().numpy().T
        got_it = its.cpu().numpy()
        for b in range(nb):
            lo, hi = b * BS, min(N, (b + 1) * BS)
            docs = CSRDocuments(ip[lo:hi + 1] - ip[lo], ii[ip[lo]:ip[hi]], cc[ip[lo]:ip[hi]])
            g, s, it = m.update_variables(docs, latents=g0[:, lo:hi], max_iter=20, return_iterations=True)
            assert np.array_equal(got_g[:, lo:hi], g) and np.array_equal(got_it[lo:hi], it), (n_ring, b)
            if n_ring == nb or b >= nb - 3:            # (a ring of three holds the last three batches)
                assert np.array_equal(ring[b % n_ring].cpu().numpy().T, s), (n_ring, b)
    go, so, ito = oracle.estep(lam, .1, (ip[:BS + 1]).astype(np.int32), ii[:ip[BS]], cc[:ip[BS]], g0[:, :BS], 20, 1e-3)
    assert relerr(got_g[:, :BS], go) < 1e-9 and np.array_equal(got_it[:BS], ito)
    # arguments: fewer than three arrays, a NULL among them
    with pytest.raises(ValueError):
        corpus_pass(m, ip.astype(np.int64), ii, cc, BS, g0_d, gam, ring[:2])
    m.close()
