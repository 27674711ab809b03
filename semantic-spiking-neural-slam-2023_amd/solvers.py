"""Build-time decoder solves: L2-regularised least squares on LIF rate curves.

``decoders = argmin_D |A D - Y|^2 + m sigma^2 |D|^2`` with ``sigma = reg * max(A)`` and ``A`` the
(m eval points) x (n neurons) activity matrix.  nengo solves the normal equations by Cholesky and
switches to the dual (m x m) system when there are fewer eval points than neurons; both forms are
implemented here.  This runs once per model build - it is not part of the step loop - so it uses
NumPy on the host for small ensembles and torch (hipBLAS/hipSOLVER on the GPU when one is visible)
for the 10^4-neuron ensembles of the benchmark configs, where ``A^T A`` alone is ~10^12 FLOP each.
"""
import numpy as np

_TORCH_MIN_WORK = 2e9   # m*n*min(m,n) above which the torch path is worth it


def lif_rates_np(J, tau_rc, tau_ref, amplitude=1.0):
    out = np.zeros_like(J)
    m = J > 1
    out[m] = amplitude / (tau_ref + tau_rc * np.log1p(1.0 / (J[m] - 1.0)))
    return out


def relu_rates_np(J, amplitude=1.0):
    return amplitude * np.maximum(J, 0.0)


def _rates_np(J, neuron):
    if neuron["type"] in ("lif", "lifrate"):
        return lif_rates_np(J, neuron["tau_rc"], neuron["tau_ref"], neuron["amplitude"])
    if neuron["type"] == "relu":
        return relu_rates_np(J, neuron["amplitude"])
    raise ValueError(neuron["type"])


class _blas_threads:
    """OpenBLAS with one thread per core is pathologically slow on the small Cholesky factorisations
    of a build (measured 133 ms vs 6 ms for 500x500 with 8 threads vs 1-4): cap it.

    Re-entrant and cheap inside an outer ``with _blas_threads():`` - the builder wraps a whole build in one, so the 8 128
    product-ensemble solves of a SLAMNetwork do not each pay threadpoolctl's scan of the loaded libraries (0.75 ms x 2 per
    solve: 19 of the 84 s of a config-3 build under cProfile, round 3)."""
    _depth = 0

    def __init__(self, n=4):
        self.n, self.ctx = n, None

    def __enter__(self):
        _blas_threads._depth += 1
        if _blas_threads._depth > 1:
            return
        try:
            from threadpoolctl import threadpool_info, threadpool_limits
            # only ever LOWER the count: a pool started with OMP_NUM_THREADS / OPENBLAS_NUM_THREADS = 1 or 2 (torchrun sets 1
            # for its workers) crashes inside dpotrs when the limit is raised past what it was started with
            cur = max([i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"] + [1])
            if cur <= self.n:
                return
            self.ctx = threadpool_limits(limits=self.n)
            self.ctx.__enter__()
        except Exception:  # pragma: no cover
            self.ctx = None

    def __exit__(self, *a):
        _blas_threads._depth -= 1
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _solve_np(A, Y, reg):
    import scipy.linalg
    m, n = A.shape
    sigma = reg * A.max()
    lam = m * sigma * sigma
    if m >= n:
        G = A.T @ A
        G[np.diag_indices(n)] += lam
        c = scipy.linalg.cho_factor(G, overwrite_a=True, check_finite=False)
        return scipy.linalg.cho_solve(c, A.T @ Y, check_finite=False)
    G = A @ A.T
    G[np.diag_indices(m)] += lam
    c = scipy.linalg.cho_factor(G, overwrite_a=True, check_finite=False)
    return A.T @ scipy.linalg.cho_solve(c, Y, check_finite=False)


def _torch_device():
    try:
        import torch
    except Exception:  # pragma: no cover
        return None, None
    return torch, (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))


def _solve_torch(eval_points, scaled_encoders, bias, neuron, Y, reg):
    torch, dev = _torch_device()
    f64 = torch.float64
    E = torch.as_tensor(scaled_encoders, dtype=f64, device=dev)
    X = torch.as_tensor(eval_points, dtype=f64, device=dev)
    J = X @ E.T + torch.as_tensor(bias, dtype=f64, device=dev)
    if neuron["type"] in ("lif", "lifrate"):
        A = torch.zeros_like(J)
        msk = J > 1
        A[msk] = neuron["amplitude"] / (neuron["tau_ref"] + neuron["tau_rc"] * torch.log1p(1.0 / (J[msk] - 1.0)))
    else:
        A = neuron["amplitude"] * torch.clamp(J, min=0.0)
    del J
    Yt = torch.as_tensor(Y, dtype=f64, device=dev)
    m, n = A.shape
    sigma = reg * float(A.max())
    lam = m * sigma * sigma
    G = A.T @ A if m >= n else A @ A.T
    G.diagonal().add_(lam)
    L = None
    for _ in range(2):                       # the Gram matrix is positive definite by construction; a failed
        L, info = torch.linalg.cholesky_ex(G)    # factorisation has only been seen with several processes sharing a GPU
        if int(info) == 0 and bool(torch.isfinite(L).all()):
            break
        L = None
        torch.cuda.synchronize() if dev.type == "cuda" else None
    if L is None:                            # last resort: host factorisation of the same system
        import scipy.linalg
        with _blas_threads():
            c = scipy.linalg.cho_factor(G.cpu().numpy(), check_finite=False)
            if m >= n:
                return scipy.linalg.cho_solve(c, (A.T @ Yt).cpu().numpy(), check_finite=False)
            return (A.T.cpu().numpy()) @ scipy.linalg.cho_solve(c, Yt.cpu().numpy(), check_finite=False)
    D = torch.cholesky_solve(A.T @ Yt, L) if m >= n else A.T @ torch.cholesky_solve(Yt, L)
    return D.cpu().numpy()


def solve_decoders(eval_points, scaled_encoders, bias, neuron, Y, reg=0.1, backend="auto"):
    """Decoders (n_neurons, size_out) for targets ``Y`` (m, size_out) at ``eval_points`` (m, dims).

    ``scaled_encoders`` is (n, dims) = ``encoders * gain / radius``; drive ``J = X E^T + bias``.
    """
    m, n = eval_points.shape[0], scaled_encoders.shape[0]
    work = float(m) * n * min(m, n)
    use_torch = backend == "torch" or (backend == "auto" and work >= _TORCH_MIN_WORK
                                       and _torch_device()[0] is not None)
    if use_torch:
        return _solve_torch(eval_points, scaled_encoders, bias, neuron, Y, reg)
    with _blas_threads():
        A = _rates_np(eval_points @ scaled_encoders.T + bias, neuron)
        return _solve_np(A, np.asarray(Y, dtype=float), reg)
