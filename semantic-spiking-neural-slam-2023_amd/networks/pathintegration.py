"""Path-integration network: one velocity-controlled oscillator (VCO) ensemble per Fourier bin.

Mirrors the reference's ``PathIntegration`` (``sspslam/networks/pathintegration.py:22-191``) - same
constructor arguments, attributes (``velocity_input``, ``input``, ``output``, ``oscillators``,
``recur_conns``, ``to_SSP``, ``to_Fourier``) and the same connection topology (checked against the
census in tests/golden/topology.json).  This module only *declares* the graph; the VCO array is the
shard unit of the hot path and is stepped by ``k_ensarray_step`` in csrc/.

State layout per VCO k (3-D ensemble, radius sqrt(2)): ``[Re F_k, Im F_k, omega_k]`` where
``omega_k = A_k . v`` is injected from ``velocity_input`` each step (synapse None) and the recurrent
connection decodes ``feedback`` (an Euler-folded limit-cycle oscillator) through ``Lowpass(tau)``.
"""
import numpy as np

from .. import frontend as nengo
from ..utils import sparsity_to_x_intercept


def _phase_tables(d):
    """cos/sin of 2*pi*(j*m mod d)/d as (d, d) tables with exact integer phase reduction."""
    jm = np.outer(np.arange(d), np.arange(d)) % d
    ang = 2.0 * np.pi * jm / d
    return np.cos(ang), np.sin(ang)


def get_to_Fourier(d):
    """(3K, d) map from an SSP to the oscillator layout, K=(d+1)//2 (reference ``:816-822``).

    Rows 3k and 3k+1 (k = 1..K-1) hold Re and Im of DFT row k; the DC rows and every omega row
    (3k+2) are zero: the DC oscillator is driven by a constant and omega comes from the velocity.
    """
    K = (d + 1) // 2
    c, s = _phase_tables(d)
    M = np.zeros((3 * K, d))
    M[3::3][:K - 1] = c[1:K]          # Re exp(-2 pi i k j / d)
    M[4::3][:K - 1] = -s[1:K]         # Im exp(-2 pi i k j / d)
    return M


def get_from_Fourier(d):
    """(d, 3K) inverse of ``get_to_Fourier`` with Hermitian completion (reference ``:824-844``).

    ``x[j] = (1/d) * sum_m Re(F[m] exp(+2 pi i j m / d))`` with ``F[m] = o[3m] + i o[3m+1]`` for
    m < K and ``F[d-m] = conj(F[m])``.  For even d the reference additionally copies the DC real
    part into the Nyquist bin (``:829-830``) - reproduced; hexagonal SSP dims are always odd.
    """
    K = (d + 1) // 2
    c, s = _phase_tables(d)
    M = np.zeros((d, 3 * K))
    m = np.arange(K)
    M[:, 3 * m] = c[:, m] / d
    M[:, 3 * m + 1] = -s[:, m] / d
    mm = np.arange(1, K)
    M[:, 3 * mm] += c[:, (d - mm) % d] / d
    M[:, 3 * mm + 1] += s[:, (d - mm) % d] / d
    if d % 2 == 0:
        M[:, 0] += c[:, d // 2] / d
        M[:, 1] = 0.0
    return M


def make_feedback(recurrent_tau, scaling_factor, length_scale0, max_radius=1.0, stable=True):
    """Decoder target of the recurrent connection (reference ``feedback`` ``:118-134``).

    ``x = [re, im, omega_scaled]``; ``w = omega_scaled / (scaling_factor * length_scale)`` is the
    angular velocity.  stable=True adds the radial attractor ``(R^2 - r^2)/r``.  Returns the
    Euler-folded next state ``tau * dx + x`` for the first two dims and 0 for the third.
    """
    tau = float(recurrent_tau)
    denom = float(scaling_factor) * float(length_scale0)
    R2 = float(max_radius) ** 2

    def feedback(x):
        w = x[2] / denom
        if stable:
            r = max(np.sqrt(x[0] * x[0] + x[1] * x[1]), 1e-9)
            g = (R2 - r * r) / r
        else:
            g = 0.0
        d0 = x[0] * g - x[1] * w
        d1 = x[1] * g + x[0] * w
        return np.array([tau * d0 + x[0], tau * d1 + x[1], 0.0])

    def feedback_batch(X):
        """Vectorised form over rows of X (used by the builder for eval points)."""
        X = np.asarray(X, dtype=float)
        w = X[:, 2] / denom
        if stable:
            r = np.maximum(np.sqrt(X[:, 0] ** 2 + X[:, 1] ** 2), 1e-9)
            g = (R2 - r * r) / r
        else:
            g = np.zeros(X.shape[0])
        out = np.zeros_like(X)
        out[:, 0] = tau * (X[:, 0] * g - X[:, 1] * w) + X[:, 0]
        out[:, 1] = tau * (X[:, 1] * g + X[:, 0] * w) + X[:, 1]
        return out

    feedback.batch = feedback_batch
    return feedback


def _identity_node_fn(t, x):
    return x


class PathIntegration(nengo.Network):
    """See module docstring; arguments as the reference (``pathintegration.py:108-111``)."""

    def __init__(self, ssp_space, n_neurons, recurrent_tau=0.05, scaling_factor=1, stable=True,
                 max_radius=1, with_gcs=False, n_gcs=1000, solver_weights=False, label="pathint",
                 **kwargs):
        super().__init__(label=label)
        if solver_weights:
            raise nengo.BuildError("solver_weights=True (full weight matrices) is not supported")
        d = ssp_space.ssp_dim
        N = ssp_space.domain_dim
        K = (d + 1) // 2
        if callable(stable):
            feedback = stable
        else:
            feedback = make_feedback(recurrent_tau, scaling_factor, ssp_space.length_scale[0, 0]
                                     if np.ndim(ssp_space.length_scale) == 2 else
                                     np.ravel(ssp_space.length_scale)[0],
                                     max_radius=max_radius, stable=bool(stable))
        self.to_SSP = get_from_Fourier(d)
        self.to_Fourier = get_to_Fourier(d)
        self.n_oscs = K
        A = ssp_space.phase_matrix

        with self:
            self.velocity_input = nengo.Node(size_in=N, label=label + "_vel_input")
            self.input = nengo.Node(size_in=d, label=label + "_input")
            if with_gcs:
                self.output = nengo.Ensemble(
                    n_gcs, d, encoders=ssp_space.sample_grid_encoders(n_gcs),
                    intercepts=nengo.Choice([sparsity_to_x_intercept(d, 0.1)]), label=label + "_output")
            else:
                self.output = nengo.Node(size_in=d, label=label + "_output")

            self.oscillators = nengo.EnsembleArray(n_neurons, K, ens_dimensions=3, radius=np.sqrt(2),
                                                   label=label + "_vco", **kwargs)
            # the reference turns the array's output node into an explicit identity function node
            self.oscillators.output.output = _identity_node_fn
            self.oscillators.output.native = ("identity",)

            nengo.Connection(self.input, self.oscillators.input, transform=self.to_Fourier)
            self.recur_conns = []
            self.vel_conns = []
            for k in range(1, K):
                ens = self.oscillators.ea_ensembles[k]
                T = np.zeros((3, N))
                T[2] = A[k]
                self.vel_conns.append(nengo.Connection(self.velocity_input, ens, transform=T, synapse=None))
                self.recur_conns.append(nengo.Connection(
                    ens, ens, function=feedback, synapse=recurrent_tau,
                    solver=nengo.LstsqL2(weights=solver_weights)))
            dc = nengo.Node([1, 0, 0], label=label + "_zerofreq")
            nengo.Connection(dc, self.oscillators.ea_ensembles[0], synapse=None)
            nengo.Connection(self.oscillators.output, self.output, transform=self.to_SSP)
