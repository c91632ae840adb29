"""Host-side and PCIe-inclusive rates around the path (never `value`): document ingestion (list
of tuples -> CSR -> device batch; text corpus -> CSR), the host-pointer E-step, whole
update_parameters calls, and the reference's README example loop.

    python tools/host_rate.py [--root TREE]      (GPU box, repo root; --root times another tree)
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--root", default=ROOT)
args = ap.parse_args()
sys.path.insert(0, args.root)
import trlda_amd                                     # noqa: E402
from trlda_amd import _ffi                           # noqa: E402
from trlda_amd.documents import CSRDocuments, DeviceBatch, as_csr   # noqa: E402
from trlda_amd.models import OnlineLDA               # noqa: E402
from trlda_amd.utils import load_documents           # noqa: E402
from trlda_amd.utils.synthetic import make_corpus    # noqa: E402

print("tree:", os.path.dirname(trlda_amd.__file__))
L = _ffi.lib()
K, V, B = 100, 7000, 200
docs = CSRDocuments(*make_corpus(B, V, seed=20150707, mean_unique=100))
lst = docs.to_list()


def rate(label, fn, n, unit_docs=B, drain=None):
    """`drain`: called before the clock starts and before it stops -- update_parameters returns
    as soon as its kernels are enqueued, so a loop without it times the host, not the calls."""
    for _ in range(12):                              # (staging buffers, device allocations, worker threads)
        fn()
    if drain:
        drain()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    if drain:
        drain()
    dt = (time.perf_counter() - t) / n
    print("%-58s %9.1f us/call -> %10.0f docs/s" % (label, dt * 1e6, unit_docs / dt))
    return dt


# ---- ingestion ---------------------------------------------------------------------------
rate("list of tuples -> CSR (as_csr)", lambda: as_csr(lst), 50)
# (round 6: trlda_batch_create validates and copies on the caller's thread and hands the index to the
# library's worker threads; a batch destroyed before anybody used it is never indexed -- so the
# "made and dropped" loop measures the caller's share only, "made, used, dropped" the whole build on
# the caller's thread (its first user takes over a build no worker has started), and "a stream of
# them" what a pipeline gets: made eight ahead of their use, indexed by the workers, each upload
# enqueued by a later trlda_batch_create on the caller's thread)


def made_used_dropped(src):
    b = DeviceBatch(src, V, 0)
    L.trlda_batch_long_word_len(b.handle)           # (the first use: waits for / builds the index)
    b.close()


def stream_of(n=400, ahead=8):
    win = [DeviceBatch(docs, V, 0) for _ in range(ahead)]
    t = time.perf_counter()
    for i in range(n):
        win.append(DeviceBatch(docs, V, 0))
        b = win.pop(0)
        L.trlda_batch_long_word_len(b.handle)
        b.close()
    dt = (time.perf_counter() - t) / n
    for b in win:
        b.close()
    return dt


rate("list of tuples -> device batch, made and dropped", lambda: DeviceBatch(lst, V, 0).close(), 50)
rate("list of tuples -> device batch, made, used, dropped", lambda: made_used_dropped(lst), 50)
rate("CSR -> device batch, made and dropped (caller's share)", lambda: DeviceBatch(docs, V, 0).close(), 50)
rate("CSR -> device batch, made, used, dropped (index on the caller)", lambda: made_used_dropped(docs), 50)
stream_of(100)
dt = stream_of()
print("%-58s %9.1f us/call -> %10.0f docs/s" % ("CSR -> device batch, a stream made 8 ahead of its use", dt * 1e6, B / dt))
big = CSRDocuments(*make_corpus(20000, V, seed=1, mean_unique=100))
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "corpus.dat")
    with open(path, "w") as f:
        for d in big.to_list():
            f.write("%d %s\n" % (len(d), " ".join("%d:%d" % t for t in d)))
    size = os.path.getsize(path) / 1e6
    try:
        from trlda_amd.utils import load_documents_csr
        rate("text (%.0f MB, 20k docs) -> CSR batches of 200" % size,
             lambda: sum(len(b) for b in load_documents_csr(path, 200)), 5, unit_docs=20000)
    except (ImportError, TypeError):
        pass
    rate("text -> lists of tuples, batches of 200 (load_documents)",
         lambda: sum(len(b) for b in load_documents(path, 200)), 2, unit_docs=20000)

# ---- the host-pointer E-step and whole updates ------------------------------------------------
L.trlda_seed(1)
m = OnlineLDA(V, K, 1000000)
g0 = np.empty((K, B), order="F")
L.trlda_sample_gamma_init(K, B, g0)
batch = m.upload(docs)


def drain():
    L.trlda_model_synchronize(m._handle)


rate("do_e_step, device-resident batch (gamma0 up, gamma + sstats down)",
     lambda: m.update_variables(batch, latents=g0, max_iter=20), 100)
rate("do_e_step, list-of-tuples docs", lambda: m.update_variables(lst, latents=g0, max_iter=20), 20)
# (every update call gets a mini-batch of its own: a model fed one mini-batch again and again fits
# it and its E-steps leave early -- 0.33 instead of 0.49 ms per call)
many = [CSRDocuments(*make_corpus(B, V, seed=777 + i, mean_unique=100)) for i in range(31)]
many_dev = [m.upload(d) for d in many]
many_lst = [d.to_list() for d in many]
for tr in (10, 0):
    for label, seq in (("device batch", many_dev), ("list of tuples", many_lst)):
        pos = [0]

        def call(seq=seq, tr=tr, pos=pos):
            m.update_parameters(seq[pos[0] % len(seq)], max_iter_tr=tr, max_iter_inference=20)
            pos[0] += 1
        dt = rate("update_parameters(max_iter_tr=%d), %s" % (tr, label), call, 30, drain=drain)

# ---- the reference's README example (README.md:36-59) on a 1000-document file, one epoch ------
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "data_train.dat")
    small = CSRDocuments(*make_corpus(1000, V, seed=5, mean_unique=100))
    with open(path, "w") as f:
        for d in small.to_list():
            f.write("%d %s\n" % (len(d), " ".join("%d:%d" % t for t in d)))
    model = OnlineLDA(num_words=7000, num_topics=100, num_documents=1000000, alpha=.1, eta=.2)

    def epoch():
        for documents in load_documents(path, 200):
            model.update_parameters(docs=documents, max_iter_tr=10, max_iter_inference=20, kappa=.7,
                                    tau=100., update_alpha=True, update_eta=True)
        model.lambdas                                    # the getter synchronises
    rate("README example, one epoch over 1000 documents (update_alpha, update_eta)", epoch, 3,
         unit_docs=1000)
    # the same loop in its steady state: 20 000 documents (100 mini-batches)
    path = os.path.join(tmp, "data_train_20k.dat")
    with open(path, "w") as f:
        for d in big.to_list():
            f.write("%d %s\n" % (len(d), " ".join("%d:%d" % t for t in d)))
    rate("README example, one epoch over 20 000 documents", epoch, 2, unit_docs=20000)

    def epoch_plain():
        for documents in load_documents(path, 200):
            model.update_parameters(docs=documents, max_iter_tr=10, max_iter_inference=20, kappa=.7, tau=100.)
        model.lambdas
    rate("the same without update_alpha / update_eta", epoch_plain, 2, unit_docs=20000)
