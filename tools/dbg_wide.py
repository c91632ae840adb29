"""debug: where does the single-orientation kernel's sstats differ (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import HipSampler, seeded_gamma, seeded_lambda
from trlda_amd import _ffi
from trlda_amd.models import OnlineLDA
from trlda_amd.documents import CSRDocuments
from oracle.pyoracle import Oracle
hip = _ffi.lib()
sampler = HipSampler(hip)
oracle = Oracle()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 333
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
V = 2500
rng = np.random.RandomState(1000 + K)
lam = seeded_lambda(sampler, 41, K, V)
lens = [0, 1, 3, 7, 8, 9, 31, 64, 79, 80, 81, 100, 127, 160, 161, 200, 255, 256, 257, 300, 420, 700, 1200]
docs, ip = [], [0]
for n in lens:
    ids = rng.permutation(V)[:n]; cnts = rng.randint(4, size=n)
    docs.append((ids, cnts)); ip.append(ip[-1] + n)
ids = np.concatenate([d[0] for d in docs]).astype(np.int32)
cnts = np.concatenate([d[1] for d in docs]).astype(np.int32)
ip = np.array(ip, np.int32)
g0 = seeded_gamma(sampler, 42, K, len(lens))
m = OnlineLDA(num_words=V, num_topics=K, num_documents=1000, alpha=.1, eta=.3); m.lambdas = lam
hip.trlda_model_set_sstats_mode(m._handle, mode)
hip.trlda_model_set_doc_kernel(m._handle, 2)
for (it, thr) in [(0, 0.), (1, 0.), (30, 1e-3)]:
    g, s, iters = m.update_variables(CSRDocuments(ip, ids, cnts), latents=g0, max_iter=it, threshold=thr, return_iterations=True)
    go, so, ito = oracle.estep(lam, .1, ip, ids, cnts, g0, it, thr)
    print("it", it, "gamma err", np.max(np.abs(g - go) / np.abs(go)), "iters eq", np.array_equal(iters, ito))
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.where(so > 0, np.abs(s - so) / so, (s != 0) * 1.0)
    bad = np.argwhere(e > 1e-9)
    print(" bad entries", len(bad), "words", np.unique(bad[:, 1])[:20], "k range", (bad[:, 0].min(), bad[:, 0].max()) if len(bad) else None)
    for w in np.unique(bad[:, 1])[:6]:
        pos = np.nonzero(ids == w)[0]
        dd = np.searchsorted(ip, pos, side="right") - 1
        print("  word", w, "docs", dd, "lens", [lens[x] for x in dd], "pos in doc", pos - ip[dd], "cnt", cnts[pos], "ratio", (s[:3, w] / so[:3, w]))
