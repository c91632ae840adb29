"""Synthetic bag-of-words corpora for tests and bench.py.

NumPy ``Generator(PCG64(seed))``; word popularity p_w ∝ 1/(w+1)^1.07 (``zipf=True``)
or uniform; a document has ``n_d = clip(1 + Poisson(mean_unique-1), 1, V)`` *unique*
word ids drawn without replacement ∝ p (Gumbel top-k == sequential
Plackett-Luce sampling) and counts ``1 + Poisson(0.6)``.  Output is the CSR form
(indptr, ids, cnts; int32) of the reference's ``vector<vector<pair<int,int>>>``
(include/lda.h:21-23).
"""
import numpy as np

SEED_BASE = 20150706


def word_popularity(V, zipf=True, exponent=1.07):
    if zipf:
        p = 1.0 / np.power(np.arange(1, V + 1, dtype=np.float64), exponent)
    else:
        p = np.ones(V, dtype=np.float64)
    return p / p.sum()


def lognormal_lengths(num_docs, seed, median=100, sigma=0.32, longest=600, p_long=0.004):
    """Heavy-tailed document lengths: log-normal around `median` (sigma 0.32: about 2 % of the
    documents have more than 192 unique words) and, with probability p_long, a document of
    `longest` / 2 .. `longest` words -- real corpora hold such documents, and the reference's own
    test_speed uses up to 600 (onlinelda_test.py:204-246)."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    n = np.rint(np.exp(np.log(median) + sigma * rng.standard_normal(num_docs)))
    long_one = rng.random(num_docs) < p_long
    n = np.where(long_one, rng.integers(longest // 2, longest + 1, size=num_docs), n)
    return np.clip(n, 1, longest).astype(np.int64)


def make_corpus(num_docs, V, seed=SEED_BASE, mean_unique=100, zipf=True, count_rate=0.6,
                chunk=2048, lengths=None):
    """-> (indptr[num_docs+1], ids[nnz], cnts[nnz]) int32 CSR.  `lengths`: the documents'
    numbers of unique words (default: 1 + Poisson(mean_unique - 1))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    logp = np.log(word_popularity(V, zipf))
    n = np.clip(1 + rng.poisson(mean_unique - 1, size=num_docs), 1, V).astype(np.int64)
    if lengths is not None:
        n = np.clip(np.asarray(lengths, dtype=np.int64), 0, V)
        if n.shape != (num_docs,):
            raise ValueError("lengths must have one entry per document")
    indptr = np.zeros(num_docs + 1, dtype=np.int64)
    np.cumsum(n, out=indptr[1:])
    ids = np.empty(indptr[-1], dtype=np.int32)
    # keep the (chunk x V) key matrix under ~256 MB
    chunk = max(1, min(chunk, (32 << 20) // max(V, 1)))
    for lo in range(0, num_docs, chunk):
        hi = min(lo + chunk, num_docs)
        keys = logp[None, :] + rng.gumbel(size=(hi - lo, V))
        kmax = max(int(n[lo:hi].max()), 1)
        top = np.argpartition(-keys, min(kmax, V - 1), axis=1)[:, :kmax] if kmax < V else \
            np.argsort(-keys, axis=1)
        # order the kmax candidates by key so the first n_d are the true top-n_d
        order = np.argsort(-np.take_along_axis(keys, top, axis=1), axis=1)
        top = np.take_along_axis(top, order, axis=1)
        for r in range(hi - lo):
            d = lo + r
            ids[indptr[d]:indptr[d + 1]] = top[r, :n[d]]
    cnts = (1 + rng.poisson(count_rate, size=indptr[-1])).astype(np.int32)
    return indptr.astype(np.int32), ids, cnts


def csr_to_docs(indptr, ids, cnts):
    """CSR -> the reference's list-of-lists-of-(id, count) tuples."""
    return [[(int(ids[i]), int(cnts[i])) for i in range(indptr[d], indptr[d + 1])]
            for d in range(len(indptr) - 1)]


def docs_to_csr(docs):
    """list-of-lists-of-(id, count) -> CSR int32 (pure NumPy helper for tests)."""
    n = np.fromiter((len(d) for d in docs), dtype=np.int64, count=len(docs))
    indptr = np.zeros(len(docs) + 1, dtype=np.int64)
    np.cumsum(n, out=indptr[1:])
    flat = np.array([t for d in docs for t in d], dtype=np.int64).reshape(-1, 2)
    return (indptr.astype(np.int32), np.ascontiguousarray(flat[:, 0], np.int32),
            np.ascontiguousarray(flat[:, 1], np.int32))
