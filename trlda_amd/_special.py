"""Host-side digamma / trigamma (NumPy, fp64) for the empirical-Bayes updates of alpha and eta
(reference src/onlinelda.cpp:116-162, which calls digamma and polygamma(1, .) =
zeta(2, .), src/utils.cpp:107-111 / src/zeta.cpp).  K- and scalar-sized work, so it stays on
the host; the K x V and K x B reductions it needs come from the device arrays."""
import numpy as np

_PSI_SERIES = (8.33333333333333333333E-2, -2.10927960927960927961E-2, 7.57575757575757575758E-3,
               -4.16666666666666666667E-3, 3.96825396825396825397E-3, -8.33333333333333333333E-3,
               8.33333333333333333333E-2)      # src/digamma.cpp:44-52


def digamma(x):
    """psi(x) for x > 0 (elementwise): upward recurrence to s >= 10, then the asymptotic series
    of src/digamma.cpp:158-171."""
    x = np.asarray(x, dtype=np.float64)
    s = x.copy()
    w = np.zeros_like(s)
    for _ in range(10):
        below = s < 10.0
        w = np.where(below, w + 1.0 / np.where(below, s, 1.0), w)
        s = np.where(below, s + 1.0, s)
    z = 1.0 / (s * s)
    p = np.full_like(z, _PSI_SERIES[0])
    for c in _PSI_SERIES[1:]:
        p = p * z + c
    return np.log(s) - 0.5 / s - z * p - w


def trigamma(x):
    """psi'(x) = zeta(2, x) for x > 0: sum_{i<m} 1/(x+i)^2 up to s = x + m >= 20, then
    1/s + 1/(2 s^2) + sum B_2k / s^(2k+1)."""
    x = np.asarray(x, dtype=np.float64)
    s = x.copy()
    w = np.zeros_like(s)
    for _ in range(20):
        below = s < 20.0
        sq = np.where(below, s, 1.0)
        w = np.where(below, w + 1.0 / (sq * sq), w)
        s = np.where(below, s + 1.0, s)
    z = 1.0 / (s * s)
    # Bernoulli numbers B2, B4, ..., B14
    series = z * (1. / 6 + z * (-1. / 30 + z * (1. / 42 + z * (-1. / 30 + z * (5. / 66 + z * (
        -691. / 2730 + z * (7. / 6)))))))
    return w + (1.0 + 0.5 / s + series) / s
