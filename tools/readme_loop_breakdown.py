"""Where the host time of the reference's README loop goes (development aid, GPU box):
loader, conversion + upload, update call, per mini-batch of 200."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trlda_amd import _ffi                           # noqa: E402
from trlda_amd.documents import CSRDocuments         # noqa: E402
from trlda_amd.models import OnlineLDA               # noqa: E402
from trlda_amd.utils import load_documents           # noqa: E402
from trlda_amd.utils.synthetic import make_corpus    # noqa: E402

V = 7000
big = CSRDocuments(*make_corpus(20000, V, seed=1, mean_unique=100))
L = _ffi.lib()
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "d.dat")
    with open(path, "w") as f:
        for d in big.to_list():
            f.write("%d %s\n" % (len(d), " ".join("%d:%d" % t for t in d)))
    model = OnlineLDA(num_words=V, num_topics=100, num_documents=1000000, alpha=.1, eta=.2)
    for eb in (False, True):
        for rep in range(2):
            t_load = t_up = t_call = 0.0
            n = 0
            t0 = time.perf_counter()
            it = iter(load_documents(path, 200))
            while True:
                a = time.perf_counter()
                try:
                    docs = next(it)
                except StopIteration:
                    break
                b = time.perf_counter()
                batch = model.upload(docs)
                c = time.perf_counter()
                model.update_parameters(batch, max_iter_tr=10, max_iter_inference=20, update_alpha=eb,
                                        update_eta=eb)
                d = time.perf_counter()
                batch.close()
                t_load += b - a
                t_up += c - b
                t_call += d - c
                n += 1
            L.trlda_model_synchronize(model._handle)
            total = time.perf_counter() - t0
        print("eb=%d: %d batches, %.0f us each: loader %.0f, upload %.0f, update call %.0f, rest %.0f" % (
            eb, n, total / n * 1e6, t_load / n * 1e6, t_up / n * 1e6, t_call / n * 1e6,
            (total - t_load - t_up - t_call) / n * 1e6))

    # the device's share: the same 100 mini-batches uploaded beforehand / one mini-batch repeated
    batches = [model.upload(d) for d in load_documents(path, 200)]
    batches = [b for b in batches if len(b)]
    for label, seq in (("100 different resident mini-batches", batches), ("one resident mini-batch", batches[:1] * 100)):
        for rep in range(2):
            L.trlda_model_synchronize(model._handle)
            t0 = time.perf_counter()
            for b in seq:
                model.update_parameters(b, max_iter_tr=10, max_iter_inference=20)
            t1 = time.perf_counter()
            L.trlda_model_synchronize(model._handle)
            t2 = time.perf_counter()
        print("%s: %.0f us per call (host returns after %.0f)" % (label, (t2 - t0) / len(seq) * 1e6,
                                                                  (t1 - t0) / len(seq) * 1e6))
