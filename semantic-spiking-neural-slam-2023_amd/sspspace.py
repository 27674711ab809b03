"""Spatial-Semantic-Pointer algebra used at build time and in the harness (host side, NumPy).

This is the host-side mirror of the reference's ``sspslam/sspspace.py`` for the parts the hot
path needs:

* ``HexagonalSSPSpace``  - phase matrix construction (reference ``sspspace.py:678-731``,
  ``conjsym`` ``:860-868``), including the ``ssp_dim`` rounding rule (``:683-686``).
* ``SSPSpace.encode``    - ``ifft(exp(i A x / l)).real`` (``:252-273``).
* ``SSPSpace.decode``    - 'from-set' grid decode (``:339-358``).
* ``get_sample_points / get_sample_pts_and_ssps`` (``:424-506``), ``make_unitary`` (``:511-514``),
  ``bind`` / ``invert`` / ``identity`` (``:520-532``), ``sample_grid_encoders`` (``:733-762``).
* ``SPSpace``            - discrete semantic pointers (``:11-182``), including the reference's
  un-renormalised Gram-Schmidt.

Same class / method names and argument meaning as the reference so harness code reads the same.
Everything here is float64 NumPy: it runs once per model build, never inside the step loop.
Golden vectors captured from the reference (tests/golden/make_golden.py) pin this file.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

__all__ = ["SPSpace", "SSPSpace", "HexagonalSSPSpace", "RandomSSPSpace", "conjsym"]


def conjsym(K):
    """Stack phases into a conjugate-symmetric (2k+1, dim) phase matrix: [0; K; -flip(K)].

    Reference: ``sspspace.py:860-868``.  Row 0 is the DC term, rows 1..k the positive
    frequencies, rows k+1..2k their mirrored negatives, so ``exp(i A x)`` is a Hermitian
    spectrum and its inverse FFT is real.
    """
    K = np.asarray(K, dtype=float)
    k = K.shape[0]
    A = np.zeros((2 * k + 1, K.shape[1]))
    A[1:k + 1] = K
    A[k + 1:] = -K[::-1]
    return A


def _unit_rows(v, eps):
    nrm = np.sqrt(np.sum(v * v, axis=-1, keepdims=True))
    return v / np.maximum(nrm, eps)


class SPSpace:
    """Discrete semantic-pointer vocabulary (reference ``sspspace.py:11-182``).

    ``vectors`` are drawn uniformly on the sphere (``RandomState(seed).randn`` rows,
    normalised - nengo's ``UniformHypersphere(surface=True).sample``), made unitary, then
    passed through the reference's sequential projection-removal loop (``:59-62``), which
    orthogonalises but does not renormalise.
    """

    def __init__(self, domain_size, dim, seed=None, vectors=None, **kwargs):
        self.domain_size = int(domain_size)
        self.dim = int(dim)
        self.rng = np.random.RandomState(seed if seed is not None else None)
        if self.domain_size == 1:
            self.vectors = np.zeros((1, self.dim))
            self.vectors[0, 0] = 1.0
        elif vectors is not None:
            self.vectors = np.array(vectors, dtype=float)
        else:
            raw = self.rng.randn(self.domain_size, self.dim)
            raw /= np.linalg.norm(raw, axis=1, keepdims=True)
            V = self.make_unitary(raw)
            for j in range(self.domain_size):
                q = V[j] / np.linalg.norm(V[j])
                if j + 1 < self.domain_size:
                    V[j + 1:] -= np.outer(V[j + 1:] @ q, q)
            self.vectors = V
        self.inverse_vectors = self.invert(self.vectors)

    def encode(self, i):
        return self.vectors[np.asarray(i).reshape(-1).astype(int)]

    def decode(self, v, **kwargs):
        return np.argmax(self.vectors @ np.atleast_2d(v).T, axis=0)

    def clean_up(self, v, **kwargs):
        return self.vectors[self.decode(v)]

    def normalize(self, v):
        return v / np.sqrt(np.sum(v ** 2))

    def make_unitary(self, v):
        fv = np.fft.fft(v, axis=1)
        return np.fft.ifft(fv / np.abs(fv), axis=1).real

    def identity(self):
        s = np.zeros(self.dim)
        s[0] = 1.0
        return s

    def bind(self, a, b):
        a, b = np.atleast_2d(a), np.atleast_2d(b)
        return np.fft.ifft(np.fft.fft(a, axis=1) * np.fft.fft(b, axis=1), axis=1).real

    def invert(self, a):
        a = np.atleast_2d(a)
        return a[:, (-np.arange(self.dim)) % self.dim]

    def get_binding_matrix(self, v):
        v = np.asarray(v).reshape(-1)
        idx = (np.arange(self.dim)[:, None] - np.arange(self.dim)[None, :]) % self.dim
        return v[idx]


class SSPSpace:
    """Continuous SSP encoder/decoder for a given phase matrix (reference ``sspspace.py:184-532``)."""

    def __init__(self, domain_dim, ssp_dim, phase_matrix, domain_bounds=None, length_scale=1,
                 rng=None):
        phase_matrix = np.asarray(phase_matrix, dtype=float)
        if phase_matrix.shape != (ssp_dim, domain_dim):
            raise ValueError(f"phase_matrix must be ({ssp_dim}, {domain_dim}), got {phase_matrix.shape}")
        if domain_bounds is not None:
            domain_bounds = np.asarray(domain_bounds, dtype=float)
            if domain_bounds.shape[0] != domain_dim:
                raise ValueError("domain_bounds must have one row per domain dimension")
        self.domain_dim = int(domain_dim)
        self.ssp_dim = int(ssp_dim)
        self.phase_matrix = phase_matrix
        self.domain_bounds = domain_bounds
        # (dim, 1) column like the reference (``:212``); PathIntegration reads length_scale[0].
        self.length_scale = np.asarray(length_scale, dtype=float) * np.ones((self.domain_dim, 1))
        self.rng = rng if rng is not None else np.random.default_rng()
        self.decoder_model = None
        self._grid_cache = {}

    # -- encoding -------------------------------------------------------------------------
    def _scaled(self, x):
        x = np.atleast_2d(np.asarray(x, dtype=float))
        return x / self.length_scale.reshape(1, -1)

    def encode_fourier(self, x):
        return np.exp(1j * (self._scaled(x) @ self.phase_matrix.T))

    def encode(self, x):
        """(num_samples, domain_dim) -> (num_samples, ssp_dim) unit-norm SSPs."""
        x = np.atleast_2d(np.asarray(x, dtype=float))
        rows = max(1, (1 << 22) // self.ssp_dim)
        if x.shape[0] <= 4 * rows:
            return np.fft.ifft(self.encode_fourier(x), axis=1).real
        # big sample grids (10^6 points in 3-D, slam.py:209): same rows, encoded a slab at a time so the
        # complex spectrum never exists for the whole grid at once
        # (slabs on a few host threads: NumPy's exp and FFT release the GIL)
        out = np.empty((x.shape[0], self.ssp_dim))

        def slab(lo):
            out[lo:lo + rows] = np.fft.ifft(self.encode_fourier(x[lo:lo + rows]), axis=1).real

        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            list(pool.map(slab, range(0, x.shape[0], rows)))
        return out

    def update_lengthscale(self, scale):
        scale = np.asarray(scale, dtype=float)
        self.length_scale = (scale * np.ones((self.domain_dim,))).reshape(-1, 1) if scale.size == 1 \
            else scale.reshape(-1, 1)
        self._grid_cache.clear()

    # -- sampling the domain --------------------------------------------------------------
    def get_sample_points(self, samples_per_dim=100, method="length-scale"):
        b = self.domain_bounds if self.domain_bounds is not None else \
            np.tile([-10.0, 10.0], (self.domain_dim, 1))
        if method == "grid":
            counts = [int(samples_per_dim)] * self.domain_dim
        elif method == "length-scale":
            counts = [2 * int(np.ceil((b[i, 1] - b[i, 0]) / self.length_scale[i, 0]))
                      for i in range(self.domain_dim)]
        else:
            raise NotImplementedError(f"sampling method {method!r} is not part of the SLAM hot path")
        axes = [np.linspace(b[i, 0], b[i, 1], counts[i]) for i in range(self.domain_dim)]
        # np.meshgrid default 'xy' indexing, flattened C-order, like the reference (``:460-464``)
        return np.stack([g.reshape(-1) for g in np.meshgrid(*axes)], axis=1)

    def get_sample_ssps(self, num_points, **kwargs):
        return self.encode(self.get_sample_points(num_points, **kwargs))

    def get_sample_pts_and_ssps(self, num_points_per_dim=100, method="grid"):
        key = (int(num_points_per_dim), method)
        if key not in self._grid_cache:
            pts = self.get_sample_points(samples_per_dim=num_points_per_dim, method=method)
            self._grid_cache = {key: (self.encode(pts), pts)}
        return self._grid_cache[key]

    def grid_factors(self, num_points_per_dim=100):
        """Factorisation of the similarities between a vector and every SSP of the sample grid
        (``get_sample_pts_and_ssps(n, 'grid')``), for the clean-up of large grids (``slam.py:209-215``).

        A grid SSP is ``ifft(prod_m exp(i A[:, m] x_m / l_m))``, so with ``X = fft(x)`` and K = (d + 1) / 2 bins

            <S_j, x> = sum_k Re( w_k conj(X_k) E1[a, k] . Erest[r, k] ),      j = a * n_rest + r,

        where ``a`` indexes axis 1 and ``r`` the remaining axes in C order (``np.meshgrid``'s 'xy' order flattened),
        w_0 = 1/d, w_k = 2/d.  Returns ``dict(dft=(2K, d), lhs=(n, 2K), rhs=(n_rest, 2K))`` of real arrays with
        (Re, Im) interleaved along the last axis - the similarities become one (n x 2K) . (2K x n_rest) product
        instead of a pass over the n^dim x d table - or None when the space does not factor this way."""
        d, dim, n = self.ssp_dim, self.domain_dim, int(num_points_per_dim)
        if dim < 2 or d % 2 == 0 or self.domain_bounds is None:
            return None
        A = np.asarray(self.phase_matrix, dtype=float)
        K = (d + 1) // 2
        if A.shape != (d, dim) or not np.allclose(A[1:], -A[1:][::-1]) or np.any(A[0] != 0):
            return None
        b = self.domain_bounds
        axes = [np.linspace(b[i, 0], b[i, 1], n) / self.length_scale[i, 0] for i in range(dim)]
        E = [np.exp(1j * np.outer(axes[m], A[:K, m])) for m in range(dim)]          # (n, K) per axis
        w = np.full(K, 2.0 / d)
        w[0] = 1.0 / d
        rest = [m for m in range(dim) if m != 1]
        Er = E[rest[0]]
        for m in rest[1:]:
            Er = (Er[:, None, :] * E[m][None, :, :]).reshape(-1, K)

        def interleave(z):
            out = np.empty(z.shape[:-1] + (2 * z.shape[-1],))
            out[..., 0::2], out[..., 1::2] = z.real, z.imag
            return out

        W = np.exp(-2j * np.pi * np.outer(np.arange(K), np.arange(d)) / d)           # X_k = sum_c x_c W[k, c]
        dft = np.empty((2 * K, d))
        dft[0::2], dft[1::2] = W.real, W.imag
        return {"dft": dft, "lhs": interleave(E[1] * w[None, :]), "rhs": interleave(Er)}

    # -- decoding --------------------------------------------------------------------------
    def decode(self, ssp, method="from-set", sampling_method="grid", num_samples=300,
               samples=None, **kwargs):
        if method != "from-set":
            raise NotImplementedError(
                f"decode method {method!r}: only 'from-set' is on the SLAM hot path (SURVEY §2 row 6)")
        ssp = np.atleast_2d(np.asarray(ssp, dtype=float))
        if samples is None:
            sample_ssps, sample_points = self.get_sample_pts_and_ssps(num_samples, sampling_method)
        else:
            sample_ssps, sample_points = samples
        nrm = np.linalg.norm(ssp, axis=1, keepdims=True)
        unit = np.where(nrm < 1e-6, ssp, ssp / np.where(nrm == 0, 1.0, nrm))
        best = np.empty(ssp.shape[0], dtype=np.int64)
        step = max(1, int(2 ** 27 // max(1, sample_ssps.shape[0])))  # bound the sims block
        for s in range(0, ssp.shape[0], step):
            best[s:s + step] = np.argmax(sample_ssps @ unit[s:s + step].T, axis=0)
        return sample_points[best]

    def clean_up(self, ssp, method="from-set", sampling_method="grid", num_samples=300):
        return self.encode(self.decode(ssp, method, sampling_method, num_samples))

    # -- algebra ---------------------------------------------------------------------------
    def normalize(self, ssp):
        return ssp / np.maximum(np.sqrt(np.sum(ssp ** 2)), 1e-8)

    def make_unitary(self, ssp):
        f = np.fft.fft(ssp)
        return np.fft.ifft(f / np.maximum(np.abs(f), 1e-8)).real

    def make_unitary_fourier(self, fssp):
        return fssp / np.maximum(np.abs(fssp), 1e-8)

    def identity(self):
        s = np.zeros(self.ssp_dim)
        s[0] = 1.0
        return s

    def bind(self, a, b):
        a, b = np.atleast_2d(a), np.atleast_2d(b)
        return np.fft.ifft(np.fft.fft(a, axis=1) * np.fft.fft(b, axis=1), axis=1).real

    def invert(self, a):
        a = np.atleast_2d(a)
        return a[:, (-np.arange(self.ssp_dim)) % self.ssp_dim]


def _simplex_phases(domain_dim):
    """(dim+1, dim) vertices of the regular simplex used by the hexagonal tiling (``:688-689``)."""
    n = domain_dim
    top = np.sqrt(1.0 + 1.0 / n) * np.eye(n) - n ** (-1.5) * (np.sqrt(n + 1.0) + 1.0)
    return np.hstack([top, n ** (-0.5) * np.ones((n, 1))]).T


class HexagonalSSPSpace(SSPSpace):
    """Hexagonal (simplex) multi-scale, multi-rotation SSP space (reference ``sspspace.py:673-731``).

    ``ssp_dim`` alone is rounded to ``2 n^2 (dim+1) + 1`` with ``n = int(sqrt((ssp_dim-1)/(2(dim+1))))``
    exactly as the reference does when ``n_rotates``/``n_scales`` keep their defaults of 5.
    For ``domain_dim >= 3`` the reference draws rotations from an *unseeded* generator
    (``:725``); pass ``rng`` (or ``phase_matrix`` via ``SSPSpace``) to pin them.  ``seed=`` is
    accepted and ignored, like the reference's ``**kwargs``.
    """

    def __init__(self, domain_dim, ssp_dim=151, n_rotates=5, n_scales=5, scale_min=1,
                 scale_max=np.pi, scale_sampling="lin", domain_bounds=None, length_scale=1,
                 rng=None, **kwargs):
        rng = rng if rng is not None else np.random.default_rng()
        if n_rotates == 5 and n_scales == 5 and ssp_dim != 151:
            n_rotates = int(np.sqrt((ssp_dim - 1) / (2 * (domain_dim + 1))))
            n_scales = n_rotates
        self.grid_basis_dim = domain_dim + 1
        self.num_grids = n_rotates * n_scales
        self.scale_min, self.scale_max = scale_min, scale_max
        self.n_scales, self.n_rotates = n_scales, n_rotates

        if domain_dim == 1:  # the reference multiplies twice (Appendix B); kept
            n_scales = n_scales * n_rotates * n_rotates
        golden = (1 + np.sqrt(5)) / 2
        if scale_sampling == "lin":
            lo = scale_max / (n_scales * (golden - 1) + 1) if scale_min is None else scale_min
            scales = np.linspace(lo, scale_max, n_scales)
        elif scale_sampling == "log":
            lo = scale_max / golden ** (n_scales - 1) if scale_min is None else scale_min
            scales = np.geomspace(lo, scale_max, n_scales)
        elif scale_sampling == "rand":
            scales = rng.uniform(0 if scale_min is None else scale_min, scale_max, n_scales)
        else:
            raise ValueError(f"unknown scale_sampling {scale_sampling!r}")

        base = _simplex_phases(domain_dim)                       # (dim+1, dim)
        scaled = np.concatenate([base * s for s in scales], 0)    # (n_scales*(dim+1), dim)
        if n_rotates == 1 or domain_dim == 1:
            phases = scaled
        else:
            if domain_dim == 2:
                th = np.linspace(0, 2 * np.pi / 3, n_rotates, endpoint=False)
                R = np.empty((n_rotates, 2, 2))
                R[:, 0, 0], R[:, 0, 1] = np.cos(th), -np.sin(th)
                R[:, 1, 0], R[:, 1, 1] = np.sin(th), np.cos(th)
            else:
                from scipy.stats import special_ortho_group
                R = special_ortho_group.rvs(domain_dim, size=n_rotates, random_state=rng)
                R = R.reshape(n_rotates, domain_dim, domain_dim)
            # rotation-major blocks: for each rotation, all scaled simplex rows rotated
            phases = np.einsum("rij,sj->rsi", R, scaled).reshape(-1, domain_dim)
        A = conjsym(phases)
        super().__init__(domain_dim, A.shape[0], A, domain_bounds=domain_bounds,
                         length_scale=length_scale, rng=rng)

    def sample_grid_encoders(self, n_neurons, method="grid", rng=None):
        """Grid-cell-like encoders: one (scale, rotation) sub-lattice per neuron (``:733-762``).

        The reference's default sampler here is 'sobol' (not on the hot path); 'grid' placement
        is supported, and the pattern assignment of the tail neurons uses ``rng``.
        """
        rng = rng if rng is not None else self.rng
        d, n, A = self.ssp_dim, self.domain_dim, self.phase_matrix
        k = (d - 1) // 2
        n_patterns = k // (n + 1)
        per_dim = int(np.ceil(n_neurons ** (1.0 / n))) if method == "grid" else n_neurons
        pts = self.get_sample_points(per_dim, method=method)[:n_neurons]
        per_pattern = n_neurons // n_patterns
        which = np.concatenate([np.repeat(np.arange(n_patterns), per_pattern),
                                rng.integers(0, n_patterns, size=n_neurons - n_patterns * per_pattern)])
        enc = np.zeros((n_neurons, d))
        for i in range(n_neurons):
            lo = 1 + which[i] * (n + 1)
            spec = np.zeros(d, dtype=complex)
            spec[lo:lo + n + 1] = np.exp(1j * A[lo:lo + n + 1] @ pts[i])
            spec[k + 1:] = np.conj(spec[1:k + 1][::-1])
            spec[0] = 1.0
            enc[i] = np.fft.ifft(spec).real
        return enc / np.linalg.norm(enc, axis=1, keepdims=True)


class RandomSSPSpace(SSPSpace):
    """Random-phase SSP space (reference ``sspspace.py:638-668``), 'norm' and 'unif' samplers."""

    def __init__(self, domain_dim, ssp_dim, domain_bounds=None, scale_min=0.25, scale_max=2.0,
                 length_scale=1, rng=None, sampler="unif", norm_scale=None, **kwargs):
        from scipy.special import gammainc
        rng = rng if rng is not None else np.random.default_rng()
        m = (ssp_dim - 1) // 2
        if sampler == "unif":
            g = rng.normal(size=(m, domain_dim))
            ssq = np.sum(g ** 2, axis=1)
            fr = scale_max * gammainc(domain_dim / 2, ssq / 2) ** (1.0 / domain_dim) / np.sqrt(ssq)
            phases = g * fr[:, None]
        elif sampler == "norm":
            if norm_scale is None:
                norm_scale = np.sqrt(np.pi / 2) * ((scale_max - scale_min) / 2 + scale_min)
            phases = rng.normal(0.0, norm_scale, size=(m, domain_dim))
        else:
            raise ValueError(f"unknown sampler {sampler!r}")
        A = conjsym(phases)
        super().__init__(domain_dim, A.shape[0], A, domain_bounds=domain_bounds,
                         length_scale=length_scale, rng=rng)


def grid_factors_from_table(S, tol=1e-9):
    """The factor tables of ``HexagonalSSPSpace.grid_factors`` recovered from a clean-up table ALONE (the reference's
    ``sample_ssps``, ``slam.py:209``: rows ``ifft(exp(i A x_j))`` over a ``np.meshgrid`` of n points per axis), for callers that
    only hold the table - the reference's own ``SLAMNetwork`` running unmodified.  With F = the half spectrum of a row,
    ``F[a * n_rest + r] = E1[a] * Erest[r]`` element-wise, so ``lhs[a] = F[a * n_rest]`` and ``rhs[r] = F[r] / F[0]`` reproduce
    every row's spectrum; n is found by trying rows = n^2, n^3, n^4 and the factorisation is verified on random rows.  Returns
    the same dict as ``grid_factors`` (factors equal up to a per-bin constant moved from one side to the other), or None."""
    S = np.asarray(S)
    rows, d = S.shape
    if d % 2 == 0 or rows < 4:
        return None
    K = (d + 1) // 2
    rs = np.random.RandomState(0)
    for dim in (2, 3, 4):
        n = int(round(rows ** (1.0 / dim)))
        if n < 2 or n ** dim != rows:
            continue
        n_rest = rows // n
        F_l = np.fft.fft(np.asarray(S[0::n_rest][:n], dtype=float), axis=1)[:, :K]           # rows a * n_rest + 0
        F_r = np.fft.fft(np.asarray(S[:n_rest], dtype=float), axis=1)[:, :K]                 # rows 0 * n_rest + r
        if np.any(np.abs(F_r[0]) < 1e-12):
            continue
        lhs, rhs = F_l, F_r / F_r[0][None, :]
        js = rs.randint(0, rows, size=min(64, rows))
        Fj = np.fft.fft(np.asarray(S[js], dtype=float), axis=1)[:, :K]
        if np.abs(lhs[js // n_rest] * rhs[js % n_rest] - Fj).max() > tol:
            continue
        # rows must be real vectors whose spectrum the half spectrum determines (Hermitian): <S_j, x> = sum_k w_k Re(conj(X_k) F_j[k])
        w = np.full(K, 2.0 / d)
        w[0] = 1.0 / d

        def interleave(z):
            out = np.empty(z.shape[:-1] + (2 * z.shape[-1],))
            out[..., 0::2], out[..., 1::2] = z.real, z.imag
            return out
        W = np.exp(-2j * np.pi * np.outer(np.arange(K), np.arange(d)) / d)
        dft = np.empty((2 * K, d))
        dft[0::2], dft[1::2] = W.real, W.imag
        return {"dft": dft, "lhs": interleave(lhs * w[None, :]), "rhs": interleave(rhs)}
    return None
