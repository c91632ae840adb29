#!/usr/bin/env python3
"""Generate tests/golden/f12_loader.npz from the REFERENCE's own text loader.

    python tests/golden/make_loader_golden.py          # needs /root/reference (this container)

The reference's ``load_documents`` (code/trlda/python/utils/load_documents.py:6-69) is plain
Python and imports under Python 3; it is loaded from where it lies, run on a corpus file this
script writes, and only its *outputs* are stored: for every case the number of documents of
each batch it yielded and the flattened (id, count) pairs -- plus the corpus text itself, which
is this script's own data.  Cases: whole file; fixed batch sizes (one that divides the number
of lines, so that the trailing batch is empty; one that does not; one larger than the file);
stochastic batch sizes under fixed NumPy seeds (including a rate small enough to draw zeros,
which make the reference yield empty batches in the middle).
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/code/trlda/python/utils/load_documents.py"

CASES = [("all", None, False, 0), ("b5", 5, False, 0), ("b6", 6, False, 0), ("b7", 7, False, 0),
         ("b100", 100, False, 0), ("s4_seed1", 4, True, 1), ("s9_seed2", 9, True, 2),
         ("s1_seed3", 1, True, 3), ("s1_seed11", 1, True, 11)]


def corpus_text(rng):
    lines = []
    for d in range(42):
        n = int(rng.integers(0, 9)) if d % 11 else 0          # some empty documents
        ids = rng.integers(0, 7000, size=n)
        cnts = rng.integers(0, 6, size=n)                     # zero counts are legal
        sep = " " if d % 5 else "\t"                          # any whitespace separates tokens
        body = sep.join("%d:%d" % (i, c) for i, c in zip(ids, cnts))
        lines.append(("%d%s%s" % (n, sep if n else "", body)) + ("  " if d % 7 == 0 else ""))
    lines[3] = "3 10:1 10:2   10:3"                           # a repeated id, runs of blanks
    lines[8] = "99 5:1"                                       # the leading count is not checked
    return "\n".join(lines) + "\n"


def flatten(batches):
    sizes = [len(b) for b in batches]
    lens = [len(d) for b in batches for d in b]
    pairs = [t for b in batches for d in b for t in d]
    arr = np.array(pairs, dtype=np.int64).reshape(-1, 2)
    return (np.array(sizes, dtype=np.int64), np.array(lens, dtype=np.int64),
            arr[:, 0].copy(), arr[:, 1].copy())


def main():
    if not os.path.exists(REF):
        sys.exit("reference not present: nothing generated")
    spec = importlib.util.spec_from_file_location("ref_load_documents", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = corpus_text(np.random.Generator(np.random.PCG64(12)))
    path = os.path.join(HERE, "_loader_corpus.tmp")
    with open(path, "w") as f:
        f.write(text)
    out = {"text": np.array(text), "cases": np.array([c[0] for c in CASES])}
    try:
        for name, batch_size, stochastic, seed in CASES:
            np.random.seed(seed)
            got = mod.load_documents(path, batch_size, stochastic)
            batches = [got] if batch_size is None else list(got)
            sizes, lens, ids, cnts = flatten(batches)
            out[name + "_sizes"], out[name + "_lens"] = sizes, lens
            out[name + "_ids"], out[name + "_cnts"] = ids, cnts
            out[name + "_args"] = np.array([batch_size or 0, int(stochastic), seed])
            # where the global NumPy stream is afterwards: the draws are part of the behaviour
            out[name + "_next"] = np.array(np.random.randint(0, 2 ** 31 - 1))
    finally:
        os.remove(path)
    np.savez_compressed(os.path.join(HERE, "f12_loader.npz"), **out)
    print("wrote f12_loader.npz:", {k: v.shape for k, v in out.items() if k.endswith("_sizes")})


if __name__ == "__main__":
    main()
