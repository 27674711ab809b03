"""Small numeric helpers the harness and network builders need.

* ``Rd_sampling``            - additive-recurrence low-discrepancy points (landmark placement);
  reference ``sspslam/utils/utils.py:41-55`` (used at ``experiments/run_slam.py:119``).
* ``sparsity_to_x_intercept`` - intercept giving a target active fraction for unit-sphere
  encoders; reference ``utils/utils.py:5-10``.
"""
import numpy as np


def _plastic_constant(d, iters=10):
    """Fixed point of ``x = (1+x)**(1/(d+1))`` reached by ``iters`` iterations from 2 (reference count)."""
    x = 2.0
    for _ in range(iters):
        x = (1.0 + x) ** (1.0 / (d + 1))
    return x


def Rd_sampling(n, d, seed=0.5):
    """``n`` points of the R_d sequence in [0,1)^d: ``frac(seed + i * alpha)``, ``alpha_j = g**-(j+1)``."""
    g = _plastic_constant(d)
    alpha = np.array([(1.0 / g) ** (j + 1) % 1 for j in range(d)])
    i = np.arange(1, n + 1, dtype=float)[:, None]
    return (seed + alpha[None, :] * i) % 1


def sparsity_to_x_intercept(d, p):
    """Intercept ``c`` such that a fraction ``p`` of unit vectors in ``d``-D has dot product > c."""
    from scipy.special import betaincinv
    sign = 1.0
    if p > 0.5:
        p, sign = 1.0 - p, -1.0
    return sign * np.sqrt(1.0 - betaincinv((d - 1) / 2.0, 0.5, 2 * p))
