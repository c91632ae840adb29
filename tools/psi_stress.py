"""Development aid (GPU box): the device psi / exp(psi) (csrc/psi.h, through trlda_debug_digamma) against
SciPy on 800 000 random arguments over 600 decades, the small integers and the switch points of
the rational recurrence.  Observed: exp(psi) within 1.9e-15 relative (per max(1, |psi|)), psi 1.6e-15."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from scipy.special import digamma
from trlda_amd import _ffi
L = _ffi.lib()
rng = np.random.RandomState(0)
x = np.concatenate([10 ** rng.uniform(-300, 300, 200000), 10 ** rng.uniform(-3, 3, 400000),
                    rng.uniform(0, 12, 200000), np.arange(1, 40, dtype=float), [1e25, 9.99e24, 1.01e25, 1e150, 1e-290, 1e-291]])
outs = [np.zeros_like(x) for _ in range(4)]
rc = L.trlda_debug_digamma(0, len(x), 0.0, x.ctypes.data, *[o.ctypes.data for o in outs])
assert rc == 0
psi, epsi, lean, em = outs
with np.errstate(all="ignore"):
    ref = digamma(x)
    eref = np.exp(ref)
ok = np.isfinite(eref) & (eref > 1e-290) & (eref < 1e290)
rel = np.abs(epsi[ok] - eref[ok]) / eref[ok] / np.maximum(1.0, np.abs(ref[ok]))
print("exp(psi): max rel err / max(1,|psi|) =", rel.max(), "at x =", x[ok][rel.argmax()], "n =", ok.sum())
perr = np.abs(psi - ref) / np.maximum(np.abs(ref), 1.0)
perr = perr[np.isfinite(perr)]
print("psi: max err =", perr.max())
print("underflow consistent:", bool(np.all(epsi[(eref == 0) & np.isfinite(ref)] == 0)))
print("positive-only form == general form (integers 1..10 apart):", bool(np.array_equal(epsi[~((x > 0) & (x <= 10) & (x == np.floor(x)))], lean[~((x > 0) & (x <= 10) & (x == np.floor(x)))])))
