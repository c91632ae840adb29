"""The update loops end to end on the GPU (reference src/onlinelda.cpp:53-111,
src/batchlda.cpp:43-61, src/cumulativelda.cpp:49-72) at BASELINE.json's big configurations,
against the pinned CPU oracle composed step by step, plus the equivalences between the fused
device path (statistics + M-step + next row sums in one kernel, active words only) and the plain
launch sequence, size-independent invariants at the full per-GPU batch sizes, and the
empirical-Bayes reductions staying on the device.
"""
import ctypes as C
import math

import numpy as np
import pytest

from helpers import TIGHT_RTOL, HipSampler, relerr, seeded_gamma

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(hip_lib):
    from trlda_amd import _ffi
    assert _ffi.device_count() >= 1, "GPU tests need a visible MI355X"
    return hip_lib


@pytest.fixture(scope="module")
def sampler(hip):
    return HipSampler(hip)


def corpus(B, V, seed, mean_unique=100):
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    return CSRDocuments(*make_corpus(B, V, seed=seed, mean_unique=mean_unique))


def random_lambda(K, V, seed):
    # (the libc-stream sampler would need K * V * 100 draws: 5e9 at K = 500, V = 100 000)
    rng = np.random.RandomState(seed)
    return np.asfortranarray(rng.gamma(100., .01, (K, V)))


def online_model(K, V, lam, D, alpha=.1, eta=.3):
    """An OnlineLDA holding `lam` without paying for the constructor's K * V * 100 draws."""
    from trlda_amd.models import OnlineLDA
    m = OnlineLDA.__new__(OnlineLDA)
    m._num_documents = int(D)
    m._update_count = 0
    m._ada_tau = 1000.
    m._ada_rho = 1. / m._ada_tau
    m._ada_sq_norm = 1.
    m._setup(V, K, alpha, eta, None, _lambda=lam)
    return m


def batch_model(K, V, lam, alpha=.1, eta=.3):
    from trlda_amd.models import BatchLDA
    m = BatchLDA.__new__(BatchLDA)
    m._setup(V, K, alpha, eta, None, _lambda=lam)
    return m


def oracle_online_update(orc, lam, alpha, eta, D, docs, g0, count, max_iter_tr, max_iter_inference,
                         kappa=.7, tau=100., nthreads=8):
    """onlinelda.cpp:53-111 composed from the oracle's pieces (its E-step on `nthreads`, which
    equals its serial one bit for bit); init_gamma=True, so g0 is the only draw."""
    B = len(docs)
    rho = math.pow(tau + count, -kappa)               # libm, as the reference (numpy.power differs by an ulp at e.g. (12, -.9))
    lam_prime = lam
    g = g0
    if max_iter_tr > 0:
        lam = orc.tr_init(lam_prime, docs.indptr, docs.ids, docs.cnts, D, rho, eta)
        for _ in range(max_iter_tr):
            g, s, _it = orc.estep(lam, alpha, docs.indptr, docs.ids, docs.cnts, g,
                                  max_iter_inference, 1e-3, nthreads=nthreads)
            lam = orc.mstep_blend(lam_prime, s, rho, eta, float(D) / B)
    else:
        g, s, _it = orc.estep(lam, alpha, docs.indptr, docs.ids, docs.cnts, g, max_iter_inference,
                              1e-3, nthreads=nthreads)
        lam = orc.mstep_blend(lam_prime, s, rho, eta, float(D) / B)
    return rho, lam, g


# --------------------------------------------------------------------------------------------
# BASELINE.json config 5: OnlineLDA K=500, V=100 000, max_iter_tr=10
# --------------------------------------------------------------------------------------------
def test_config5_online_update_trust_region(hip, oracle, sampler):
    import trlda_amd
    K, V, B, D = 500, 100000, 256, 1000000
    lam0 = random_lambda(K, V, 5)
    m = online_model(K, V, lam0, D)
    lam = lam0
    for call, seed in enumerate((31, 32)):           # the 2nd call starts from carried row sums
        docs = corpus(B, V, seed=500 + call)
        trlda_amd.seed(seed)
        rho = m.update_parameters(docs, max_iter_tr=10, max_iter_inference=20)
        g0 = seeded_gamma(sampler, seed, K, B)
        rho_o, lam, _g = oracle_online_update(oracle, lam, .1, .3, D, docs, g0, call, 10, 20)
        assert rho == rho_o
        got = m.lambdas
        assert relerr(got, lam) < TIGHT_RTOL, (call, relerr(got, lam))
    assert m.update_count == 2


@pytest.mark.parametrize("K,V,B", [(200, 25000, 96), (500, 20000, 40), (384, 20000, 200)])
def test_big_table_m_step_leaves_exp_psi_lambda_behind(hip, oracle, sampler, monkeypatch, K, V, B):
    """K > 128 on a big table: from the second trust-region iteration on the M-step kernel has left
    exp(psi(lambda)) of the batch's words behind and the single-orientation document kernel applies
    the topic factors (estep_docs_wide_kernel<KS, true>) -- no exp_elog_beta_kernel.  Against the
    oracle, against the same calls with the switch off (TRLDA_BIG_EMIT=0), and the path is the one
    that ran: trlda_model_last_preamble_fused."""
    import trlda_amd
    D = 200000
    lam0 = random_lambda(K, V, 9)
    docs = [corpus(B, V, seed=900 + i) for i in range(2)]
    got = {}
    for switch in ("1", "0"):
        monkeypatch.setenv("TRLDA_BIG_EMIT", switch)
        m = online_model(K, V, lam0, D)
        for i, (tr, seed) in enumerate(((3, 51), (2, 52))):
            trlda_amd.seed(seed)
            m.update_parameters(docs[i], max_iter_tr=tr, max_iter_inference=20)
            # (taken from K = 257 on: below that the factor multiply costs the document kernel more
            # than the separate kernel takes)
            assert hip.trlda_model_last_preamble_fused(m._handle) == (int(switch) if K > 256 else 0)
        got[switch] = m.lambdas
        m.close()
    lam = lam0
    for i, (tr, seed) in enumerate(((3, 51), (2, 52))):
        g0 = seeded_gamma(sampler, seed, K, B)
        _r, lam, _g = oracle_online_update(oracle, lam, .1, .3, D, docs[i], g0, i, tr, 20)
    assert relerr(got["1"], lam) < TIGHT_RTOL, relerr(got["1"], lam)
    assert relerr(got["0"], lam) < TIGHT_RTOL
    assert relerr(got["1"], got["0"]) < 1e-10


def test_config5_single_step_and_plain_sequence(hip, oracle, sampler):
    """max_iter_tr=0 (onlinelda.cpp:103-109: the in-place M-step on the active words plus the
    decay of the others), and the plain launch sequence (fused update and carried row sums off)
    on the same inputs: both equal the oracle, and each other to rounding."""
    import trlda_amd
    K, V, B, D = 500, 100000, 128, 500000
    lam0 = random_lambda(K, V, 6)
    docs = [corpus(B, V, seed=600 + i) for i in range(2)]
    results = []
    for fused in (1, 0):
        m = online_model(K, V, lam0, D)
        hip.trlda_model_set_fused_update(m._handle, fused)
        hip.trlda_model_set_carry_rowsums(m._handle, fused)
        for i, (tr, seed) in enumerate(((0, 41), (2, 42))):
            trlda_amd.seed(seed)
            m.update_parameters(docs[i], max_iter_tr=tr, max_iter_inference=20)
        results.append(m.lambdas)
        m.close()
    lam = lam0
    for i, (tr, seed) in enumerate(((0, 41), (2, 42))):
        g0 = seeded_gamma(sampler, seed, K, B)
        _r, lam, _g = oracle_online_update(oracle, lam, .1, .3, D, docs[i], g0, i, tr, 20)
    assert relerr(results[0], lam) < TIGHT_RTOL
    assert relerr(results[1], lam) < TIGHT_RTOL
    assert relerr(results[0], results[1]) < 1e-12


# --------------------------------------------------------------------------------------------
# BASELINE.json config 4: BatchLDA K=200, V=50 000, max_iter_inference=100 (its default)
# --------------------------------------------------------------------------------------------
def test_config4_batch_update(hip, oracle, sampler):
    import trlda_amd
    K, V, B = 200, 50000, 512
    lam0 = random_lambda(K, V, 4)
    docs = corpus(B, V, seed=400)
    m = batch_model(K, V, lam0)
    trlda_amd.seed(51)
    m.update_parameters(docs, max_epochs=2, max_iter_inference=100)
    sampler.seed(51)
    lam = lam0
    for _epoch in range(2):                          # batchlda.cpp:48-61
        g0 = sampler.sample_gamma(K, B, 100) / 100.
        _g, s, _it = oracle.estep(lam, .1, docs.indptr, docs.ids, docs.cnts, g0, 100, 1e-3, nthreads=8)
        lam = .3 + s
    got = m.lambdas
    assert relerr(got, lam) < TIGHT_RTOL, relerr(got, lam)
    untouched = np.setdiff1d(np.arange(V), docs.ids)
    assert np.all(got[:, untouched] == .3)           # lambda = eta where the batch has no word


# --------------------------------------------------------------------------------------------
# full per-GPU batch sizes: size-independent properties (SURVEY.md a17 carried through the M-step)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,V,B,tr", [(500, 100000, 512, 10), (500, 100000, 4096, 2),
                                      (100, 7000, 1600, 10)])
def test_online_update_mass_balance_full_batch(hip, K, V, B, tr):
    """sum(sstats) = sum(counts) (a17), so after any number of trust-region M-steps
    sum(lambda) = (1-rho) sum(lambda') + rho (K V eta + D/B sum(counts)); and the words outside
    the batch hold exactly (1-rho) lambda' + rho eta."""
    import trlda_amd
    D = 1000000
    lam0 = random_lambda(K, V, K + B)
    docs = corpus(B, V, seed=B)
    m = online_model(K, V, lam0, D)
    trlda_amd.seed(7)
    rho = m.update_parameters(docs, max_iter_tr=tr, max_iter_inference=20)
    got = m.lambdas
    total = float(docs.cnts.sum())
    want = (1. - rho) * lam0.sum() + rho * (K * V * .3 + D / float(B) * total)
    assert abs(got.sum() - want) < 1e-9 * want
    untouched = np.setdiff1d(np.arange(V), docs.ids)
    if len(untouched):
        assert relerr(got[:, untouched], (1. - rho) * lam0[:, untouched] + rho * .3) < 1e-14
    assert np.isfinite(got).all() and (got > 0).all()


def test_batch_update_mass_balance_full_batch(hip):
    """BatchLDA at config 4's per-GPU batch: sum(lambda) = K V eta + sum(counts)."""
    import trlda_amd
    K, V, B = 200, 50000, 12500
    lam0 = random_lambda(K, V, 44)
    docs = corpus(B, V, seed=44)
    m = batch_model(K, V, lam0)
    trlda_amd.seed(8)
    m.update_parameters(docs, max_epochs=1, max_iter_inference=100)
    got = m.lambdas
    total = float(docs.cnts.sum())
    want = K * V * .3 + total
    assert abs(got.sum() - want) < 1e-9 * want


# --------------------------------------------------------------------------------------------
# fused device path == plain launch sequence, small tables and odd shapes
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,V,B", [(100, 7000, 200), (7, 900, 33), (333, 13000, 40),
                                   (129, 40000, 64), (512, 9000, 24),
                                   (100, 1500, 3000), (200, 2000, 2500), (129, 1200, 2000)])
def test_fused_update_equals_plain_sequence(hip, oracle, sampler, K, V, B):
    """Every update loop through both device paths: OnlineLDA with and without the trust region
    (twice, so that the second call runs on carried row sums), BatchLDA, CumulativeLDA.  K odd
    (8-byte streaming accesses), K = 512 (four topic blocks), V not a multiple of anything; and
    thousands of documents over a small vocabulary: word lists of hundreds of entries walked by
    single wavefronts (the statistics kernels' per-batch threshold, kLongWord), the 512 longest by
    whole workgroups."""
    import trlda_amd
    from trlda_amd.models import BatchLDA, CumulativeLDA
    D = 20000
    lam0 = random_lambda(K, V, K)
    docs = [corpus(B, V, seed=K + i, mean_unique=min(60, V // 4)) for i in range(3)]
    if B >= 2000:
        from trlda_amd.documents import DeviceBatch
        dev = DeviceBatch(docs[0], V, 0)
        assert hip.trlda_batch_long_word_len(dev.handle) >= 64
        assert 0 < hip.trlda_batch_num_long_words(dev.handle) <= 512
        dev.close()
    out = {}
    for fused in (1, 0):
        m = online_model(K, V, lam0, D)
        hip.trlda_model_set_fused_update(m._handle, fused)
        hip.trlda_model_set_carry_rowsums(m._handle, fused)
        trlda_amd.seed(61)
        m.update_parameters(docs[0], max_iter_tr=3, max_iter_inference=20)
        m.update_parameters(docs[1], max_iter_tr=0, max_iter_inference=20)
        m.update_parameters(docs[2], max_iter_tr=2, max_iter_inference=20, init_gamma=False)
        out["online", fused] = m.lambdas
        m.close()
        trlda_amd.seed(62)
        b = BatchLDA(num_words=V, num_topics=K) if K * V < 2000000 else batch_model(K, V, lam0)
        hip.trlda_model_set_fused_update(b._handle, fused)
        hip.trlda_model_set_carry_rowsums(b._handle, fused)
        trlda_amd.seed(63)
        b.update_parameters(docs[0], max_epochs=3, max_iter_inference=30)
        out["batch", fused] = b.lambdas
        b.close()
        if K * V < 2000000:
            trlda_amd.seed(64)
            c = CumulativeLDA(num_words=V, num_topics=K)
            hip.trlda_model_set_fused_update(c._handle, fused)
            hip.trlda_model_set_carry_rowsums(c._handle, fused)
            c.update_parameters(docs[0], max_epochs=2, max_iter_inference=30)
            c.update_parameters(docs[1], max_epochs=1, max_iter_inference=30)
            out["cumulative", fused] = c.lambdas
            c.close()
    for kind in ("online", "batch", "cumulative"):
        if (kind, 1) in out:
            # (round 4: the fused path adds its row sums up in another order again -- merged launch,
            # list segments: a few 1e-12 after three calls)
            assert relerr(out[kind, 1], out[kind, 0]) < 5e-11, (kind, relerr(out[kind, 1], out[kind, 0]))
    # and the fused online trajectory against the oracle
    lam = lam0
    sampler.seed(61)
    for i, tr in enumerate((3, 0, 2)):
        rho = math.pow(100. + i, -.7)
        lam_prime = lam
        if tr > 0:
            lam = oracle.tr_init(lam_prime, docs[i].indptr, docs[i].ids, docs[i].cnts, D, rho, .3)
        g = None
        for j in range(max(tr, 1)):
            if g is None or i == 2:                  # init_gamma=False on the third call
                g = sampler.sample_gamma(K, B, 100) / 100.
            g, s, _it = oracle.estep(lam, .1, docs[i].indptr, docs[i].ids, docs[i].cnts, g, 20, 1e-3,
                                     nthreads=8)
            lam = oracle.mstep_blend(lam_prime, s, rho, .3, float(D) / B)
    assert relerr(out["online", 1], lam) < TIGHT_RTOL


def test_tiny_row_sums_do_not_take_the_fused_preamble(hip, oracle):
    """Row sums below ~1.4e-3 make exp(-psi(row sum)) overflow, which the fused small-table
    preamble would multiply into an underflowed exp(psi(lambda)): the host keeps a lower bound of
    the row sums and uses the two-kernel preamble there (the reference's single exponential)."""
    K, V, B = 16, 40, 12
    rng = np.random.RandomState(3)
    lam = np.asfortranarray(rng.uniform(1e-6, 3e-6, (K, V)))        # row sums ~8e-5
    lam[3, :] = rng.uniform(.5, 1.5, V)                             # one ordinary topic
    docs = corpus(B, V, seed=9, mean_unique=10)
    g0 = np.asfortranarray(rng.gamma(100., .01, (K, B)))
    m = online_model(K, V, lam, 100, alpha=.1, eta=1e-6)
    g, s = m.update_variables(docs, latents=g0, max_iter=20)
    assert hip.trlda_model_last_preamble_fused(m._handle) == 0
    go, so, _ = oracle.estep(lam, .1, docs.indptr, docs.ids, docs.cnts, g0, 20, 1e-3)
    assert np.isfinite(g).all() and np.isfinite(s).all()
    assert relerr(g, go) < 1e-8
    big = so > 1e-290
    assert relerr(s[big], so[big]) < 1e-8
    # an ordinary lambda on the same model takes the fused preamble again
    m.lambdas = random_lambda(K, V, 1)
    m.update_variables(docs, latents=g0, max_iter=5)
    assert hip.trlda_model_last_preamble_fused(m._handle) == 1
    # ... and stays on it through updates (the bound follows the M-step)
    m.update_parameters(docs, max_iter_tr=2)
    m.update_variables(docs, latents=g0, max_iter=5)
    assert hip.trlda_model_last_preamble_fused(m._handle) == 1


# --------------------------------------------------------------------------------------------
# empirical Bayes / adaptive rate: K-sized traffic only
# --------------------------------------------------------------------------------------------
def test_empirical_bayes_moves_only_k_sized_results(hip):
    import trlda_amd
    K, V, B, D = 100, 7000, 200, 100000
    m = online_model(K, V, random_lambda(K, V, 2), D)
    docs = m.upload(corpus(B, V, seed=21))
    trlda_amd.seed(5)
    before = hip.trlda_model_d2h_bytes(m._handle)
    for _ in range(3):
        m.update_parameters(docs, max_iter_tr=2, update_alpha=True, update_eta=True, adaptive=True)
    m.update_parameters(docs, max_iter_tr=0, update_lambda=False, update_alpha=True, update_eta=True)
    moved = hip.trlda_model_d2h_bytes(m._handle) - before
    assert moved < 4 * (3 * K * 8 + 3 * 4096 * 8 + 2048 * 8), moved     # << K * V * 8 = 5.6 MB
    assert moved < K * V * 8 // 20
    assert np.isfinite(m.alpha).all() and m.eta > 0


def test_device_eb_statistics_match_numpy(hip):
    """The three reductions against NumPy on downloaded copies (K = 200: two topics per thread
    in the gamma kernel; B not a multiple of the chunk)."""
    import trlda_amd
    from scipy import special as _special       # (the checker: the product's psi is in C)
    K, V, B, D = 200, 3000, 77, 5000
    lam0 = random_lambda(K, V, 12)
    m = online_model(K, V, lam0, D)
    docs = m.upload(corpus(B, V, seed=22, mean_unique=50))
    hip.trlda_model_set_keep_sstats(m._handle, 1)
    trlda_amd.seed(9)
    gamma = np.empty((K, B), order="F")
    count, rho_out = C.c_int(0), C.c_double(0.)
    from trlda_amd import _ffi
    _ffi.check(hip.trlda_model_online_update(m._handle, docs.handle, D, .3, 2, 20, .7, 100., -1., 1, 1,
                                             0.001, C.byref(count), C.byref(rho_out),
                                             gamma.ctypes.data))
    want = (_special.digamma(gamma) - _special.digamma(gamma.sum(axis=0))[None, :]).sum(axis=1)
    assert relerr(m._psi_gamma_diff_device(B), want) < 1e-11
    lam = m.lambdas
    total, rowsums = m._lambda_psi_stats_device()
    assert relerr(rowsums, lam.sum(axis=1)) < 1e-13
    assert abs(total - _special.digamma(lam).sum()) < 1e-11 * abs(total)
    sstats = np.empty((K, V), order="F")
    _ffi.check(hip.trlda_model_get_sstats(m._handle, sstats))
    upd = (.3 + D / float(B) * sstats) - lam0
    u2, g2 = C.c_double(0.), C.c_double(0.)
    _ffi.check(hip.trlda_model_adaptive_stats(m._handle, .3, D / float(B), 1000., C.byref(u2),
                                              C.byref(g2)))
    assert abs(u2.value - (upd * upd).sum()) < 1e-11 * u2.value
    grad = upd / 1000.
    assert abs(g2.value - (grad * grad).sum()) < 1e-11 * g2.value


# --------------------------------------------------------------------------------------------
# multi-GPU composition in C over a real RCCL communicator (world size 1 on this box)
# --------------------------------------------------------------------------------------------
def test_online_update_multi_over_rccl_world_1(hip, tmp_path):
    """trlda_model_online_update_multi with an ncclComm_t made by ncclCommInitRank (one rank):
    E-step -> ncclAllReduce of the statistics on the model's stream -> M-step, no Python in
    between; equals the single-GPU update.  In a child process: the communicator stays there."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_c_abi.py"
    script.write_text('''
import ctypes as C, os, sys
sys.path.insert(0, %r)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch                                   # brings librccl.so into the process
import trlda_amd
from trlda_amd import _ffi
from trlda_amd.documents import CSRDocuments
from trlda_amd.models import OnlineLDA
from trlda_amd.utils.synthetic import make_corpus
rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
uid, comm = UniqueId(), C.c_void_p()
torch.cuda.set_device(0)
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
L = _ffi.lib()
K, V, D, B = 24, 900, 5000, 60
docs = [CSRDocuments(*make_corpus(B, V, seed=70 + i, mean_unique=40)) for i in range(2)]
trlda_amd.seed(4)
a = OnlineLDA(num_words=V, num_topics=K, num_documents=D)
ra = [a.update_parameters(d, max_iter_tr=tr) for d, tr in zip(docs, (3, 0))]
trlda_amd.seed(4)
b = OnlineLDA(num_words=V, num_topics=K, num_documents=D)
count, rho, rb = C.c_int(0), C.c_double(0.), []
for d, tr in zip(docs, (3, 0)):
    batch = b.upload(d)
    _ffi.check(L.trlda_model_online_update_multi(b._handle, batch.handle, comm, B, 0, D, .3, tr, 20,
                                                 .7, 100., -1., 1, 0.001, C.byref(count), C.byref(rho)))
    rb.append(rho.value)
assert ra == rb and count.value == 2, (ra, rb, count.value)
err = float(np.max(np.abs(a.lambdas - b.lambdas) / a.lambdas))
assert err < 1e-10, err
# the bare all-reduce: world 1 leaves the buffer unchanged
buf = torch.arange(K * V, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()                       # the model runs on a stream of its own
_ffi.check(L.trlda_model_allreduce_sstats(b._handle, comm, C.c_void_p(buf.data_ptr())))
_ffi.check(L.trlda_model_synchronize(b._handle))
assert torch.equal(buf.cpu(), torch.arange(K * V, dtype=torch.float64))
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("RCCL-C-ABI-OK")
''' % root)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
    assert "RCCL-C-ABI-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


# --------------------------------------------------------------------------------------------
# sampleGamma on the device, from the host's libc stream
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols,passes", [(100, 200, 100), (7, 3, 100), (13, 77, 31), (1, 5, 4),
                                              (500, 999, 100), (64, 4, 1)])
def test_device_gamma_draw_is_the_libc_stream(hip, rows, cols, passes):
    """rng_kernels.h against the host draw (tests/test_boundary.py pins THAT one bit for bit
    against the reference): the same integers in the same order, hence the same uniforms; the device
    takes ONE logarithm per block of 25 passes (of the product of their |u|, csrc/rng_kernels.h)
    where the host adds up 100 -- 1.4e-15 apart at most over 2 * 10^5 elements in a NumPy model of
    both, the device's form the one closer to the exact sum; and the host generator ends up exactly
    where the host draw leaves it."""
    import trlda_amd
    from trlda_amd import _ffi
    m = online_model(4, 16, random_lambda(4, 16, 1), 10)
    dev = _ffi.vp()
    n = rows * cols
    _ffi.check(hip.trlda_dev_alloc(0, n * 8, C.byref(dev)))
    got = np.empty((rows, cols), order="F")
    for seed in (42, 7):
        trlda_amd.seed(seed)
        want = np.empty((rows, cols), order="F")
        hip.trlda_sample_gamma(rows, cols, passes, want)
        after_host = np.empty((3, 3), order="F")
        hip.trlda_sample_gamma(3, 3, 2, after_host)
        trlda_amd.seed(seed)
        _ffi.check(hip.trlda_model_sample_gamma(m._handle, rows, cols, passes, 1., dev))
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        _ffi.check(hip.trlda_dev_download(0, got.ctypes.data, dev, n * 8))
        after_dev = np.empty((3, 3), order="F")
        hip.trlda_sample_gamma(3, 3, 2, after_dev)
        assert relerr(got, want) < 4e-15, relerr(got, want)
        assert np.array_equal(after_dev, after_host)
    hip.trlda_dev_free(0, dev)


@pytest.mark.parametrize("rows,cols,lo,hi,passes", [(100, 200, 0, 200, 100), (7, 3, 0, 3, 100), (13, 77, 0, 77, 31),
                                                     (37, 53, 7, 30, 100), (37, 53, 30, 53, 128), (100, 224, 0, 224, 100)])
def test_device_gamma_draw_summed_where_it_is_formed(hip, rows, cols, lo, hi, passes, monkeypatch):
    """draw_sum_kernel (logarithms summed in the workgroup that forms them, windows in segment-major
    order) against draw_abs_kernel + gamma_sum_kernel: the same products and logarithms in the same order --
    bitwise equal, whole matrices and a data-parallel rank's columns, element counts that are and are
    not multiples of the segment length."""
    import trlda_amd
    from trlda_amd import _ffi
    m = online_model(4, 16, random_lambda(4, 16, 1), 10)
    n = rows * (hi - lo)
    dev = _ffi.vp()
    _ffi.check(hip.trlda_dev_alloc(0, n * 8, C.byref(dev)))
    got = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("TRLDA_RNG_FUSED", fused)
        trlda_amd.seed(31)
        _ffi.check(hip.trlda_model_sample_gamma_cols(m._handle, rows, cols, lo, hi, passes, 100., dev))
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        got[fused] = np.empty((rows, hi - lo), order="F")
        _ffi.check(hip.trlda_dev_download(0, got[fused].ctypes.data, dev, n * 8))
    hip.trlda_dev_free(0, dev)
    assert np.array_equal(got["0"], got["1"])
    assert np.isfinite(got["1"]).all() and (got["1"] > 0).all()


def test_device_draw_keeps_the_early_exits_at_baseline_size(hip):
    """The device's gamma0 differs from the host's in the last bits (<= 4e-15: one logarithm per 25
    passes, test_device_gamma_draw_is_the_libc_stream): at
    BASELINE's headline size that must not flip a document's early exit (lda.cpp:202-203) --
    eight mini-batches of 200 documents at K = 100, V = 7000 on a peaked lambda, 300 iterations
    allowed: iteration counts equal document by document, gamma within 1e-10; and eight
    update_parameters calls with the device draw against the host draw end at the same lambda."""
    import trlda_amd
    from trlda_amd import _ffi
    K, V, B, D = 100, 7000, 200, 1000000
    lam0 = np.asfortranarray(np.random.RandomState(71).gamma(.3, 1., (K, V)) + .01)
    m = online_model(K, V, lam0, D)
    dev = _ffi.vp()
    _ffi.check(hip.trlda_dev_alloc(0, K * B * 8, C.byref(dev)))
    exits = 0
    for i in range(8):
        docs = corpus(B, V, seed=700 + i, mean_unique=100)
        trlda_amd.seed(900 + i)
        g_host = np.empty((K, B), order="F")
        hip.trlda_sample_gamma_init(K, B, g_host)
        trlda_amd.seed(900 + i)
        _ffi.check(hip.trlda_model_sample_gamma(m._handle, K, B, 100, 100., dev))
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        g_dev = np.empty((K, B), order="F")
        _ffi.check(hip.trlda_dev_download(0, g_dev.ctypes.data, dev, K * B * 8))
        assert relerr(g_dev, g_host) < 4e-15
        a = m.update_variables(docs, latents=g_host, max_iter=300, return_iterations=True)
        b = m.update_variables(docs, latents=g_dev, max_iter=300, return_iterations=True)
        assert np.array_equal(a[2], b[2]), i
        assert relerr(b[0], a[0]) < 1e-10
        exits += int((a[2] < 300).sum())
    assert exits > 100                                       # early exits did occur
    hip.trlda_dev_free(0, dev)
    out = []
    for host in (0, 1):
        mm = online_model(K, V, lam0, D)
        hip.trlda_model_set_host_gamma_draw(mm._handle, host)
        trlda_amd.seed(77)
        for i in range(8):
            mm.update_parameters(corpus(B, V, seed=700 + i, mean_unique=100), max_iter_tr=2,
                                 max_iter_inference=50)
        out.append(mm.lambdas)
    # (the last-bit differences in gamma0 grow through 8 x 2 E-steps on a peaked lambda: 2e-9
    # observed; the bar of BASELINE.json is 1e-5)
    assert relerr(out[0], out[1]) < 1e-7


def test_gamma_drawn_ahead_keeps_the_order_of_draws(hip, oracle):
    """The next update's gamma0 is drawn ahead on a stream of its own with the host stream
    advanced before its turn (trlda_model_set_draw_ahead): whatever touches the generator in
    between -- a host draw, a seed, another model, another batch size, a lower bound -- the
    ORDER of draws stays the reference's: the same sequence of calls with drawing ahead on and
    off ends at bitwise the same lambda, gamma and generator state, for OnlineLDA (with and
    without the trust-region loop), BatchLDA epochs and CumulativeLDA."""
    import trlda_amd
    from trlda_amd.models import BatchLDA, CumulativeLDA
    K, V, D = 60, 1500, 20000
    lam0 = random_lambda(K, V, 91)
    batches = [corpus(B, V, seed=950 + i, mean_unique=40) for i, B in enumerate((70, 70, 70, 33, 70, 70))]

    def run(ahead):
        out = []
        trlda_amd.seed(123)
        m = online_model(K, V, lam0, D)
        other = online_model(K, V, lam0, D)
        for mm in (m, other):
            assert hip.trlda_model_set_draw_ahead(mm._handle, ahead) == 0
        m.update_parameters(batches[0], max_iter_tr=0)            # draws, then draws ahead for 70
        m.update_parameters(batches[1], max_iter_tr=3)            # claims it
        g = np.empty((4, 5), order="F")
        hip.trlda_sample_gamma_init(4, 5, g)                      # a host draw in between
        out.append(g.copy())
        m.update_parameters(batches[2], max_iter_tr=2, init_gamma=False)   # a draw per iteration
        m.update_parameters(batches[3], max_iter_tr=0)            # another batch size
        other.update_parameters(batches[4], max_iter_tr=1)        # another model
        out.append(m.lower_bound(batches[4]))                     # draws its own gamma (host)
        m.update_parameters(batches[5], max_iter_tr=2, update_alpha=True, update_eta=True)
        trlda_amd.seed(5)                                         # a seed with a draw ahead pending
        m.update_parameters(batches[0], max_iter_tr=0)
        out += [m.lambdas, other.lambdas, m.alpha.copy(), np.array(m.eta)]
        b = BatchLDA.__new__(BatchLDA)
        b._setup(V, K, .1, .3, None, _lambda=lam0)
        hip.trlda_model_set_draw_ahead(b._handle, ahead)
        b.update_parameters(batches[0], max_epochs=3, max_iter_inference=30)
        out.append(b.lambdas)
        c = CumulativeLDA(V, K)
        hip.trlda_model_set_draw_ahead(c._handle, ahead)
        c.update_parameters(batches[1], max_epochs=2, max_iter_inference=30)
        c.update_parameters(batches[2], max_epochs=2, max_iter_inference=30)
        out.append(c.lambdas)
        state = np.zeros(33, dtype=np.uint32)
        hip.trlda_rng_get_state(state)
        tail = np.empty((3, 3), order="F")
        hip.trlda_sample_gamma_init(3, 3, tail)
        out.append(tail)
        for mm in (m, other, b, c):
            mm.close()
        return out

    on, off = run(1), run(0)
    for a, b in zip(on, off):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    # ... and drawn ahead INSIDE the call's document launch (mode 2, the default since round 6:
    # csrc/rng_kernels.h, aux_draw_workgroup): the same sequence, bitwise
    inside = run(2)
    for a, b in zip(inside, off):
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("K,V,B", [(100, 7000, 200), (60, 900, 33), (128, 3000, 224), (2, 300, 5), (100, 2000, 97)])
def test_gamma_drawn_inside_the_document_launch(hip, K, V, B):
    """Round 6: extra workgroups of an update call's merged document launch draw the NEXT fresh
    gamma0 (lda.cpp:135) -- per workgroup its own chunks of 124 elements, windows by jump matrices
    of the chunk stride and of K * B, the walk in registers.  Against every draw in its turn: bitwise
    the same gamma0 (read back through update trajectories), lambda and generator state, with and
    without the trust-region loop, with a draw per iteration, and with a batch of another size in
    between (the speculation is dropped and repeated in its turn); and the launches did carry draws."""
    import trlda_amd
    D = 50000
    lam0 = random_lambda(K, V, 17)
    batches = [corpus(B, V, seed=400 + i, mean_unique=min(60, V // 6)) for i in range(6)]
    odd = corpus(max(1, B // 2), V, seed=399, mean_unique=min(60, V // 6))

    def run(mode):
        trlda_amd.seed(321)
        m = online_model(K, V, lam0, D)
        assert hip.trlda_model_set_draw_ahead(m._handle, mode) == 0
        out = []
        m.update_parameters(batches[0], max_iter_tr=0, max_iter_inference=20)
        m.update_parameters(batches[1], max_iter_tr=0, max_iter_inference=20)
        out.append(m.lambdas)
        m.update_parameters(batches[2], max_iter_tr=3, max_iter_inference=20)
        m.update_parameters(batches[3], max_iter_tr=2, max_iter_inference=20, init_gamma=False)
        out.append(m.lambdas)
        m.update_parameters(odd, max_iter_tr=0, max_iter_inference=20)      # another shape: not claimed
        m.update_parameters(batches[4], max_iter_tr=1, max_iter_inference=20)
        g, _ = m.update_variables(batches[5], max_iter=10)                    # a fresh gamma0 of the same shape
        out += [m.lambdas, g]
        drawn = hip.trlda_model_inlaunch_draws(m._handle)
        state = np.zeros(33, dtype=np.uint32)
        hip.trlda_rng_get_state(state)
        out.append(state)
        m.close()
        return out, drawn

    (inside, n_in), (turn, n_turn) = run(2), run(0)
    assert n_turn == 0
    if B == 224:
        # 28 free CUs, 232 chunks: nine a workgroup -- drawn in its turn; only the launch of the
        # half-sized batch carries a draw (for ITS shape: never claimed)
        assert n_in == 1, n_in
    else:
        assert n_in >= 5, n_in              # (the launches of this shape carry them)
    for a, b in zip(inside, turn):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_host_gamma_draw_switch_gives_the_same_update(hip):
    import trlda_amd
    K, V, B, D = 50, 2000, 70, 10000
    lam0 = random_lambda(K, V, 3)
    docs = corpus(B, V, seed=33, mean_unique=50)
    out = []
    for host in (0, 1):
        m = online_model(K, V, lam0, D)
        hip.trlda_model_set_host_gamma_draw(m._handle, host)
        trlda_amd.seed(17)
        m.update_parameters(docs, max_iter_tr=2, init_gamma=False)
        m.update_parameters(docs, max_iter_tr=0)
        out.append(m.lambdas)
        m.close()
    assert relerr(out[0], out[1]) < 1e-11


def test_device_gamma_draw_of_a_column_range(hip):
    """A data-parallel rank draws only its own documents' columns of the mini-batch's gamma0 and
    stays in step with the stream: the columns equal the host matrix's, the generator ends where
    the full draw ends."""
    import trlda_amd
    from trlda_amd import _ffi
    K, B = 37, 53
    m = online_model(4, 16, random_lambda(4, 16, 1), 10)
    trlda_amd.seed(99)
    want = np.empty((K, B), order="F")
    hip.trlda_sample_gamma_init(K, B, want)
    after_host = np.empty((2, 2), order="F")
    hip.trlda_sample_gamma(2, 2, 3, after_host)
    for lo, hi in ((0, B), (0, 7), (7, 30), (30, B), (11, 11)):
        trlda_amd.seed(99)
        n = max(hi - lo, 1) * K
        dev = _ffi.vp()
        _ffi.check(hip.trlda_dev_alloc(0, n * 8, C.byref(dev)))
        _ffi.check(hip.trlda_model_sample_gamma_cols(m._handle, K, B, lo, hi, 100, 100., dev))
        _ffi.check(hip.trlda_model_synchronize(m._handle))
        got = np.empty((K, max(hi - lo, 1)), order="F")
        _ffi.check(hip.trlda_dev_download(0, got.ctypes.data, dev, n * 8))
        hip.trlda_dev_free(0, dev)
        if hi > lo:
            assert relerr(got[:, :hi - lo], want[:, lo:hi]) < 4e-15, (lo, hi)
        after = np.empty((2, 2), order="F")
        hip.trlda_sample_gamma(2, 2, 3, after)
        assert np.array_equal(after, after_host), (lo, hi)


# --------------------------------------------------------------------------------------------
# the next batch's preamble as extra workgroups of this call's document-kernel launch
# --------------------------------------------------------------------------------------------
def test_announced_next_batch_changes_nothing(hip, oracle):
    """trlda_model_estep_io_next over a stream of batches -- right announcements, wrong ones, a
    batch announced and then destroyed, lambda replaced and updated in between, long documents
    that leave the register tier -- against the same calls without announcements: bitwise the
    same gamma, statistics and iteration counts; and against the oracle."""
    import trlda_amd
    from trlda_amd import _ffi
    from trlda_amd.documents import CSRDocuments
    from trlda_amd.utils.synthetic import make_corpus
    K, V, B, D = 100, 7000, 200, 100000
    lams = [random_lambda(K, V, 80), random_lambda(K, V, 81)]
    csrs = [corpus(B, V, seed=800 + i) for i in range(4)]
    csrs.append(CSRDocuments(*make_corpus(40, V, seed=9, mean_unique=400)))    # beyond 192 words
    rng = np.random.RandomState(8)
    g0s = [np.asfortranarray(rng.gamma(100., .01, (K, len(c)))) for c in csrs]

    def run(announce):
        trlda_amd.seed(5)                                # the update below draws its gamma0
        m = online_model(K, V, lams[0], D)
        dev = [m.upload(c) for c in csrs]
        bufs = []
        for c, g in zip(csrs, g0s):
            ptrs = [_ffi.vp() for _ in range(4)]
            n = K * len(c)
            for p, nbytes in zip(ptrs, (n * 8, n * 8, K * V * 8, len(c) * 4)):
                _ffi.check(hip.trlda_dev_alloc(0, nbytes, C.byref(p)))
            _ffi.check(hip.trlda_dev_upload(0, ptrs[0], g.ctypes.data, n * 8))
            bufs.append(ptrs)
        out = []

        def estep(i, nxt):
            g0d, gd, sd, itd = bufs[i]
            _ffi.check(hip.trlda_model_estep_io_next(
                m._handle, dev[i].handle, dev[nxt].handle if (announce and nxt is not None) else None,
                g0d, gd, sd, 20, 1e-3, itd))
            _ffi.check(hip.trlda_model_synchronize(m._handle))
            g = np.empty((K, len(csrs[i])), order="F")
            s = np.empty((K, V), order="F")
            it = np.empty(len(csrs[i]), dtype=np.int32)
            _ffi.check(hip.trlda_dev_download(0, g.ctypes.data, gd, g.nbytes))
            _ffi.check(hip.trlda_dev_download(0, s.ctypes.data, sd, s.nbytes))
            _ffi.check(hip.trlda_dev_download(0, it.ctypes.data, itd, it.nbytes))
            out.append((g, s, it))

        for i, nxt in ((0, 1), (1, 2), (2, 0), (0, 3), (1, 1), (1, 4), (4, 0), (0, None)):
            estep(i, nxt)                                # (0, 3) then batch 1: a wrong announcement
        m.lambdas = lams[1]
        estep(2, 3)
        estep(3, 0)
        m.update_parameters(dev[1], max_iter_tr=2)       # lambda written after batch 0 was announced
        estep(0, 2)
        estep(2, 1)
        tmp = m.upload(csrs[3])                          # announce a batch, destroy it, run another
        g0d, gd, sd, itd = bufs[1]
        _ffi.check(hip.trlda_model_estep_io_next(m._handle, dev[1].handle,
                                                 tmp.handle if announce else None, g0d, gd, sd, 20, 1e-3, itd))
        tmp.close()
        estep(3, None)
        out.append((m.lambdas, np.zeros(1), np.zeros(1)))
        for ptrs in bufs:
            for p in ptrs:
                hip.trlda_dev_free(0, p)
        m.close()
        return out

    with_next, without = run(True), run(False)
    for n, (a, b) in enumerate(zip(with_next, without)):
        for q, (x, y) in enumerate(zip(a, b)):
            assert np.array_equal(x, y), (n, q, float(np.max(np.abs(x - y))))
    go, so, ito = oracle.estep(lams[0], .1, csrs[1].indptr, csrs[1].ids, csrs[1].cnts, g0s[1], 20, 1e-3,
                               nthreads=8)
    g, s, it = with_next[1]                              # batch 1, prepared under batch 0's launch
    assert relerr(g, go) < TIGHT_RTOL and np.array_equal(it, ito)
    assert relerr(s[so > 0], so[so > 0]) < TIGHT_RTOL


def test_next_preamble_from_the_m_step_changes_nothing_but_rounding(hip):
    """Small tables: the M-step kernel leaves exp(psi(lambda)) and grouped row sums for the next
    E-step (two launches per trust-region iteration); with trlda_model_set_next_preamble(0) every
    E-step launches its preamble.  Same exp(psi) values, row sums added in a different (fixed)
    order: lambda agrees to rounding, for OnlineLDA with the trust region, BatchLDA epochs and
    CumulativeLDA epochs; and repeating a run reproduces it bit for bit."""
    import trlda_amd
    from trlda_amd.models import BatchLDA, CumulativeLDA
    K, V, B, D = 100, 5000, 150, 40000
    lam0 = random_lambda(K, V, 77)
    docs = [corpus(B, V, seed=900 + i, mean_unique=80) for i in range(2)]

    def run(emit):
        out = []
        m = online_model(K, V, lam0, D)
        hip.trlda_model_set_next_preamble(m._handle, emit)
        trlda_amd.seed(71)
        m.update_parameters(docs[0], max_iter_tr=5, max_iter_inference=20)
        m.update_parameters(docs[1], max_iter_tr=4, max_iter_inference=20, init_gamma=False)
        out.append(m.lambdas)
        m.close()
        b = batch_model(K, V, lam0)
        hip.trlda_model_set_next_preamble(b._handle, emit)
        trlda_amd.seed(72)
        b.update_parameters(docs[0], max_epochs=4, max_iter_inference=25)
        out.append(b.lambdas)
        b.close()
        trlda_amd.seed(73)
        c = CumulativeLDA(num_words=V, num_topics=K)
        hip.trlda_model_set_next_preamble(c._handle, emit)
        c.update_parameters(docs[0], max_epochs=3, max_iter_inference=25)
        out.append(c.lambdas)
        c.close()
        return out

    on, off, again = run(1), run(0), run(1)
    for a, b, c in zip(on, off, again):
        assert relerr(a, b) < 1e-12, relerr(a, b)
        assert np.array_equal(a, c)


def test_word_count_sums_beyond_32_bits(hip, oracle, sampler):
    """The trust-region initial step (onlinelda.cpp:79-86) takes the words' count sums from the
    batch (formed on the host, int32); a sum that does not fit 32 bits sends the call to the
    device's own sums.  Both against the oracle."""
    import trlda_amd
    from trlda_amd.documents import CSRDocuments
    K, V, D = 16, 200, 5000
    lam0 = random_lambda(K, V, 88)
    for big in (1000, 2000000000):                   # 3 x 2e9 for word 7: 6e9 > 2^31
        indptr = np.array([0, 3, 5, 8], dtype=np.int32)
        ids = np.array([7, 11, 150, 7, 60, 7, 11, 199], dtype=np.int32)
        cnts = np.array([big, 2, 1, big, 3, big, 1, 4], dtype=np.int32)
        docs = CSRDocuments(indptr, ids, cnts)
        m = online_model(K, V, lam0, D)
        trlda_amd.seed(91)
        rho = m.update_parameters(docs, max_iter_tr=2, max_iter_inference=20)
        g0 = seeded_gamma(sampler, 91, K, 3)
        rho_o, lam, _g = oracle_online_update(oracle, lam0, .1, .3, D, docs, g0, 0, 2, 20)
        assert rho == rho_o
        assert relerr(m.lambdas, lam) < 1e-8, (big, relerr(m.lambdas, lam))
        m.close()


def test_deferred_empirical_bayes_step_changes_nothing(hip):
    """update_parameters(update_alpha, update_eta) leaves the wait for the device sums to the next
    call (the host prepares the next mini-batch meanwhile): same alpha, eta and lambda, bit for
    bit, as finishing the step at once; the getters finish it; between the halves the C ABI
    refuses E-steps and a new alpha."""
    import trlda_amd
    K, V, B, D = 100, 7000, 200, 100000
    lam0 = random_lambda(K, V, 9)
    batches = [corpus(B, V, seed=880 + i).to_list() for i in range(3)]
    results = []
    for at_once in (False, True):
        m = online_model(K, V, lam0, D)
        for i, docs in enumerate(batches):
            trlda_amd.seed(70 + i)
            m.update_parameters(docs, max_iter_tr=2, max_iter_inference=10, update_alpha=True,
                                update_eta=True)
            if at_once:
                m._settle()
            else:
                assert hip.trlda_model_online_eb_pending(m._handle) == 1
        if not at_once:
            # the C ABI between the halves
            dev = m.upload(batches[0])
            assert hip.trlda_model_estep_resident(m._handle, dev.handle, 5, 1e-3) == trlda_amd._ffi.ERR_ARG
            assert hip.trlda_model_set_alpha(m._handle, np.full(K, .1)) == trlda_amd._ffi.ERR_ARG
            dev.close()
            eta = m.eta                                          # finishes the step
            assert hip.trlda_model_online_eb_pending(m._handle) == 0
            results.append((m.alpha, eta, m.lambdas))
        else:
            results.append((m.alpha, m.eta, m.lambdas))
        m.close()
    (a0, e0, l0), (a1, e1, l1) = results
    assert np.array_equal(a0, a1) and e0 == e1 and np.array_equal(l0, l1)
    assert np.all(a0 != .1) and e0 != .3


@pytest.mark.parametrize("K,V,B", [(100, 7000, 200), (60, 900, 33), (128, 3000, 150), (2, 300, 5), (64, 20000, 100)])
def test_decay_of_the_inactive_words_inside_the_document_launch(hip, K, V, B):
    """Round 6: in update_parameters(max_iter_tr=0) (onlinelda.cpp:103-109) the words outside the
    mini-batch decay, lambda = (1 - rho) lambda + rho eta, by auxiliary workgroups of the document
    launch -- inactive_update_stream_kernel's 1024-thread blocks gone through by 512 threads, the same
    per-thread sums in the same slot order -- instead of a kernel behind it: bitwise the same lambda
    after every call of a chain (the chain also carries the row sums into the next E-step), the
    same gamma of an E-step on the result; and the launches did carry the pass."""
    import trlda_amd
    D = 30000
    lam0 = random_lambda(K, V, 23)
    batches = [corpus(B, V, seed=600 + i, mean_unique=min(60, V // 6)) for i in range(5)]
    res = []
    for aux in (1, 0):
        trlda_amd.seed(55)
        m = online_model(K, V, lam0, D)
        assert hip.trlda_model_set_aux_decay(m._handle, aux) == 0
        out = []
        for i in range(4):
            m.update_parameters(batches[i], max_iter_tr=0, max_iter_inference=20)
            out.append(m.lambdas)
        m.update_parameters(batches[4], max_iter_tr=2, max_iter_inference=20)
        m.update_parameters(batches[0], max_iter_tr=0, max_iter_inference=20, rho=1.)   # a = 0: not carried
        out.append(m.lambdas)
        g0 = np.asfortranarray(np.random.RandomState(3).gamma(1., 1., (K, B)))
        out.append(m.update_variables(batches[1], latents=g0, max_iter=10)[0])
        n = hip.trlda_model_inlaunch_decays(m._handle)
        assert (n >= 4) if aux else (n == 0), (aux, n)
        res.append(out)
        m.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    # against the oracle's M-step on untouched words: lambda = (1 - rho) lambda0 + rho eta after call 1
    ids = np.unique(batches[0].ids)
    untouched = np.setdiff1d(np.arange(V), ids)
    rho = 100. ** -.7                                          # tau = 100, kappa = .7, the first update
    if untouched.size:
        assert relerr(res[0][0][:, untouched], (1. - rho) * lam0[:, untouched] + rho * .3) < 1e-14
