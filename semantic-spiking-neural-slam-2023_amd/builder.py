"""Network -> BuiltModel: the frozen, flat operator list both the HIP backend and the oracle run.

nengo separates *build* (sampling + least squares -> arrays) from *step* (arithmetic on those
arrays).  This module is the build half for the object kinds the SLAM networks use:
it samples ensemble parameters (Appendix A.3), solves decoders (A.5), lowers every Connection to
operators on one flat signal vector, merges the thousands of per-ensemble operators of an
``EnsembleArray`` into a handful of wide ones (nengo's operator-merge pass, A.12) and orders them
by nengo's per-signal rule *sets -> incs -> reads -> updates* (A.1).

Signals live in one flat vector, laid out in arenas so that a single fill resets every
accumulator:  R (reset to 0 each step: node / ensemble inputs, neuron currents) | W (written each
step: decoded "weighted" outputs, function-node outputs, spikes) | S (persistent synapse states) |
C (constants) | T (tabulated t-only node outputs).

Operator kinds (dict ``kind``):
  fill      sig[dst:dst+len] = value
  table     sig[dst:dst+width] = rows[idx[step]]           (t-only Node, pre-tabulated per run)
  axpy      dst (+)= alpha * src                           (mode "inc" | "set")
  matvec    dst (+)= W[rows x cols] @ src                  (W is a buffer; mode "inc" | "set")
  lowpass   dst = a*dst + (1-a)*gain*src                   (Appendix A.6, an *update*)
  ensarray  K equal ensembles: J = bias + enc.x ; neuron step ; decoded rows -> sig[dst_idx]
  neurons   neuron step on a current vector J -> spike vector
  pes / voja / cleanup / gate                              (SLAM; Appendix A.7, A.8, slam.py:212-237)
"""
import math
import time

import numpy as np

from . import frontend as fe
from .solvers import solve_decoders, _blas_threads
from .glue import collapse_glue
from .stages import stage_ops

MICRO_KINDS = ("fill", "table", "axpy", "lowpass", "matvec_small", "gate")


# --------------------------------------------------------------------------------------------
class Ref:
    """A contiguous range inside one arena (resolved to an absolute offset at finalisation)."""
    __slots__ = ("arena", "off", "len")

    def __init__(self, arena, off, length):
        self.arena, self.off, self.len = arena, int(off), int(length)

    def slice(self, start, length):
        if start < 0 or start + length > self.len:
            raise fe.BuildError(f"slice [{start}:{start + length}] outside signal of size {self.len}")
        return Ref(self.arena, self.off + start, length)

    def __repr__(self):
        return f"{self.arena}[{self.off}:{self.off + self.len}]"


class BuiltEnsemble:
    """``sim.data[ens]`` (run_slam.py:266): the sampled / derived parameters of one ensemble."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class BuiltConnection:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class BuiltModel:
    def __init__(self, dt):
        self.dt = float(dt)
        self.sig_size = 0
        self.sig_init = None
        self.arena_base = {}
        self.buffers = []        # np.ndarray (float64 or int32)
        self.buffer_meta = []    # {"name", "role": param|state|learned|table|index}
        self.ops = []
        self.probes = []         # {"probe", "src", "width", "every"} | {"probe","buf",...}
        self.tables = []         # {"node","fn","width","rows_buf","idx_buf"}
        self.params = {}         # frontend object -> Built*
        self.sig = {}            # (obj, "in"|"out"|...) -> (abs_off, len)
        self.stats = {}
        self.label = None

    def add_buffer(self, arr, name, role="param"):
        self.buffers.append(arr)
        self.buffer_meta.append({"name": name, "role": role})
        return len(self.buffers) - 1

    @property
    def n_neurons(self):
        return self.stats.get("n_neurons", 0)


# --------------------------------------------------------------------------------------------
def _contig(view_or_obj):
    """(object, start, length) for an object or a contiguous ObjView."""
    if hasattr(view_or_obj, "obj") and hasattr(view_or_obj, "slice"):
        if hasattr(view_or_obj, "indices"):
            idx = np.asarray(view_or_obj.indices)
        else:  # foreign ObjView
            size = max(view_or_obj.obj.size_in, view_or_obj.obj.size_out)
            idx = np.atleast_1d(np.arange(size)[view_or_obj.slice])
        if idx.size == 0 or np.any(np.diff(idx) != 1):
            raise fe.BuildError(f"only contiguous ascending slices are supported, got {view_or_obj!r}")
        return view_or_obj.obj, int(idx[0]), int(idx.size)
    return view_or_obj, 0, None


def _kind(obj):
    n = type(obj).__name__
    if isinstance(obj, fe.Node) or n == "Node":
        return "node"
    if isinstance(obj, fe.Ensemble) or n == "Ensemble":
        return "ensemble"
    if isinstance(obj, fe.Neurons) or n == "Neurons":
        return "neurons"
    if isinstance(obj, fe.LearningRule) or n == "LearningRule":
        return "rule"
    if isinstance(obj, fe.Connection) or n == "Connection":
        return "connection"
    raise fe.BuildError(f"unsupported object {obj!r}")


def _neuron_desc(nt):
    n = type(nt).__name__
    if n == "LIF":
        return dict(type="lif", tau_rc=float(nt.tau_rc), tau_ref=float(nt.tau_ref),
                    min_voltage=float(getattr(nt, "min_voltage", 0.0)), amplitude=float(getattr(nt, "amplitude", 1.0)))
    if n == "LIFRate":
        return dict(type="lifrate", tau_rc=float(nt.tau_rc), tau_ref=float(nt.tau_ref), min_voltage=0.0,
                    amplitude=float(getattr(nt, "amplitude", 1.0)))
    if n == "RectifiedLinear":
        return dict(type="relu", tau_rc=0.0, tau_ref=0.0, min_voltage=0.0, amplitude=float(getattr(nt, "amplitude", 1.0)))
    raise fe.BuildError(f"unsupported neuron type {nt!r} (lif, lifrate, relu are available)")


def _sample(dist_or_vals, n, d, rng, default):
    v = default if (dist_or_vals is fe.Default or type(dist_or_vals).__name__ == "DefaultType") else dist_or_vals
    if hasattr(v, "sample"):
        return np.asarray(v.sample(n, d, rng=rng) if d is not None else v.sample(n, rng=rng), dtype=float)
    return np.array(v, dtype=float)


def _lowpass_coeff(tau, dt):
    return math.exp(-dt / tau) if tau > 0 else 0.0


# --------------------------------------------------------------------------------------------
DFT_KINDS = {("A", False): 1, ("B", False): 2, ("A", True): 3, ("B", True): 4}


def dft_structure(T):
    """0, or the code of the real-DFT map a transform matrix is: 1-4 = ``transform_in(d, 'A'|'B', invert)``
    (reference ``binding.py:23-54``), 5 = ``transform_out(d)`` (``:57-74``) - the three dense matrices of a
    CircularConvolution network."""
    from .networks.binding import transform_in, transform_out
    T = np.asarray(T)
    if T.ndim != 2:
        return 0
    r, c = T.shape
    if c >= 8 and r == 4 * (c // 2 + 1):
        for (align, inv), code in DFT_KINDS.items():
            if np.allclose(T, transform_in(c, align, inv), rtol=0, atol=1e-12):
                return code
    if r >= 8 and c == 4 * (r // 2 + 1) and np.allclose(T, transform_out(r), rtol=0, atol=1e-12):
        return 5
    return 0


class Builder:
    def __init__(self, network, dt=0.001, seed=None, n_eval_points=None, solver_backend="auto",
                 vco_shard=None, progress=None, probes=None, prune=False, staged=True, neuron_shard=None, replicate=(),
                 collapse=True):
        self.net, self.dt = network, float(dt)
        self.n_eval_points = n_eval_points
        self.solver_backend = solver_backend
        self.vco_shard = vco_shard           # (rank, world): build only this rank's slice of every EnsembleArray
        # (rank, world): the rank's share of EVERY neuron population (SLAMNetwork): EnsembleArrays are split over
        # ensembles, dense ensembles over neurons (rows of the encoders, columns of the decoders / PES matrix); every
        # decoded vector is then a partial sum, completed by ONE all-reduce per timestep (shard_phases below).  Ensembles
        # in `replicate` are kept whole on every rank (their decoded output reaches other neurons within the timestep).
        self.neuron_shard = neuron_shard
        self.replicated = {id(e) for e in replicate}
        if neuron_shard is not None:
            if vco_shard is not None:
                raise fe.BuildError("vco_shard and neuron_shard are alternatives")
            self.vco_shard = neuron_shard      # EnsembleArrays: the same split over ensembles
            staged = False                     # (every operator in the per-timestep core: the exchange cuts the timestep in two)
        self.progress = progress
        self.r_allocs = []
        self.staged = staged and neuron_shard is None
        self.probes_override = probes        # build with these probes instead of the network's own
        self.prune = prune                   # drop operators that no probe (transitively) depends on
        self.collapse = collapse             # fold the linear glue of the per-timestep core into lincomb operators (glue.py)
        self.model = BuiltModel(dt)
        self.arena_size = {a: 0 for a in "RWSCT"}
        self.inits = []                      # (Ref, values)
        self.raw_ops = []
        self.seed_of = {}
        root = getattr(network, "seed", None)
        if root is None:
            root = seed if seed is not None else np.random.randint(2 ** 31 - 1)
        self.root_seed = int(root)

    # -- allocation ------------------------------------------------------------------------
    def alloc(self, arena, size, init=None):
        r = Ref(arena, self.arena_size[arena], size)
        if arena == "R" and size:
            self.r_allocs.append((r.off, int(size)))
        self.arena_size[arena] += int(size)
        if init is not None:
            self.inits.append((r, np.asarray(init, dtype=float).reshape(-1)))
        return r

    def op(self, kind, **kw):
        kw["kind"] = kind
        kw["seq"] = len(self.raw_ops)
        self.raw_ops.append(kw)
        return kw

    # -- build -----------------------------------------------------------------------------
    def build(self):
        with _blas_threads():              # one BLAS-thread cap for the whole build (see solvers._blas_threads)
            return self._build()

    def _build(self):
        t0 = time.time()
        net = self.net
        ensembles = list(net.all_ensembles)
        nodes = list(net.all_nodes)
        conns = list(net.all_connections)
        probes = list(net.all_probes)
        rng = np.random.RandomState(self.root_seed)
        all_probes = probes
        if self.probes_override is not None:
            probes = list(self.probes_override)
        for obj in ensembles + nodes + conns + all_probes + [p for p in probes if p not in all_probes]:
            s = getattr(obj, "seed", None)
            self.seed_of[id(obj)] = int(s) if s is not None else int(rng.randint(2 ** 31 - 1))

        # ensemble-array blocks (equal ensembles stepped by one kernel)
        self.block_of = {}
        self.blocks = []
        for sub in ([net] + list(net.all_networks)):
            eas = getattr(sub, "ea_ensembles", None)
            if not eas:
                continue
            e0 = eas[0]
            same = all(e.n_neurons == e0.n_neurons and e.dimensions == e0.dimensions
                       and _neuron_desc(e.neuron_type) == _neuron_desc(e0.neuron_type) for e in eas)
            if not same:
                continue
            lo, hi = 0, len(eas)
            if self.vco_shard is not None:
                rank, world = self.vco_shard
                per = -(-len(eas) // world)
                lo, hi = min(len(eas), rank * per), min(len(eas), (rank + 1) * per)
            blk = {"net": sub, "ens": list(eas), "rows": [[] for _ in eas], "range": (lo, hi)}
            self.blocks.append(blk)
            for i, e in enumerate(eas):
                self.block_of[id(e)] = (blk, i)

        # signals for ensembles and nodes
        self.ens_in, self.ens_J, self.ens_spk, self.node_in, self.node_out = {}, {}, {}, {}, {}
        self.dense_rows = {}
        for blk in self.blocks:
            e0 = blk["ens"][0]
            blk["x"] = self.alloc("R", len(blk["ens"]) * e0.dimensions)
            for i, e in enumerate(blk["ens"]):
                self.ens_in[id(e)] = blk["x"].slice(i * e0.dimensions, e0.dimensions)
        for e in ensembles:
            if id(e) in self.block_of:
                continue
            self.ens_in[id(e)] = self.alloc("R", e.dimensions)
            self.ens_J[id(e)] = self.alloc("R", self._local_count(e))
            self.ens_spk[id(e)] = self.alloc("W", self._local_count(e))
            self.dense_rows[id(e)] = []
        for n in nodes:
            self._alloc_node(n)

        self.built_ens = {}
        self.rule_in = {}
        self.learned = {}
        self.pending = {}        # id(ensemble) -> [(conn-like, pre_start, pre_len, transform, callback)]
        # parameters of every ensemble (cheap, vectorised); decoders are solved on demand
        for e in ensembles:
            self._build_ensemble_params(e)
        conns = self._drop_dead_ends(conns, nodes, probes)
        for c in conns:
            self._lower_connection(c)
        for p in probes:
            self._lower_probe(p)
        for e in ensembles:
            self._solve_pending(e)
        for e in ensembles:
            if id(e) not in self.block_of:
                self._emit_dense_ensemble(e)
        self._live = self._live_elements(probes)
        for blk in self.blocks:
            self._emit_block(blk)
        self._finalise()
        self.model.stats["build_seconds"] = time.time() - t0
        self.model.stats["n_neurons"] = int(sum(e.n_neurons for e in ensembles))
        return self.model

    def _drop_dead_ends(self, conns, nodes, probes):
        """Connections into passthrough Nodes that nothing reads (no outgoing connection, no probe) -
        e.g. the unused default ``output`` of the squaring EnsembleArrays (binding.py:297-317) - cannot
        influence any result; skipping them saves their decoder solves and per-step decoding."""
        probed = {id(_contig(p.target)[0]) for p in probes}
        conns = list(conns)
        while True:
            has_out = {id(_contig(c.pre)[0]) for c in conns}
            dead = {id(n) for n in nodes if getattr(n, "output", None) is None and n.size_in > 0
                    and id(n) not in has_out and id(n) not in probed}
            kept = [c for c in conns if id(_contig(c.post)[0]) not in dead]
            if len(kept) == len(conns):
                return kept
            conns = kept

    # -- nodes -----------------------------------------------------------------------------
    def _alloc_node(self, n):
        out = getattr(n, "output", None)
        if out is None:
            r = self.alloc("R", n.size_in)
            self.node_in[id(n)] = r
            self.node_out[id(n)] = r
        elif callable(out) and n.size_in == 0:
            r = self.alloc("T", n.size_out)
            self.node_out[id(n)] = r
            self.model.tables.append({"node": n, "fn": out, "width": n.size_out, "ref": r})
            self.op("table", dst=r, width=n.size_out, table=len(self.model.tables) - 1)
        elif callable(out):
            rin = self.alloc("R", n.size_in)
            rout = self.alloc("W", n.size_out)
            self.node_in[id(n)], self.node_out[id(n)] = rin, rout
            native = getattr(n, "native", None)
            if native is None:
                native = self._probe_function_node(n)
            if native[0] == "identity":
                self.op("axpy", dst=rout, src=rin, len=n.size_in, alpha=1.0, mode="set")
            elif native[0] == "cleanup":
                table = np.ascontiguousarray(native[1], dtype=float)
                b = self.model.add_buffer(table, f"cleanup_table_{len(self.model.buffers)}")
                extra = {}
                gf = native[2] if len(native) > 2 else None
                if gf is not None and gf["lhs"].shape[0] * gf["rhs"].shape[0] == table.shape[0]:
                    # factor tables of the sample grid (sspspace.grid_factors): the device may form the
                    # similarities as a small matrix product instead of a pass over the whole table
                    extra = {"g_" + nm: self.model.add_buffer(np.ascontiguousarray(gf[nm], dtype=float),
                                                              f"cleanup_{nm}_{len(self.model.buffers)}")
                             for nm in ("dft", "lhs", "rhs")}
                    extra.update(grid_rows=int(gf["lhs"].shape[0]), grid_cols=int(gf["rhs"].shape[0]),
                                 grid_k2=int(gf["lhs"].shape[1]))
                self.op("cleanup", dst=rout, src=rin, rows=table.shape[0], cols=table.shape[1], w=b, **extra)
            elif native[0] == "gate":
                _, d, thres, rate = native
                self.op("gate", dst=rout, src=rin, d=int(d), thres=float(thres), rate=float(rate))
            else:
                raise fe.BuildError(f"unknown native node op {native[0]!r}")
        else:
            vals = np.asarray(out, dtype=float).reshape(-1)
            self.node_out[id(n)] = self.alloc("C", vals.size, init=vals)

    def _probe_function_node(self, n):
        """A Python function node with inputs must map to a kernel.  The three such nodes of the reference's networks are
        recognised from their BEHAVIOUR (random probes), with candidates for their constants taken from the function's closure
        cells - so the reference's classes run with zero edits (no ``node.native`` hints):

        * the identity (``EnsembleArray.output`` / passthrough lambdas, reference ``pathintegration.py:167``);
        * the clean-up ``lambda t, x: clean_up_fun(x)`` = ``S[argmax(S @ x)]`` (``slam.py:212-215,270``): S is a 2-D array
          with ``size_in`` columns found in the closure; every probe must return exactly the row the table product selects;
        * the gate ``update_state_func`` (``slam.py:233-237,249``): ``rate * (a - b)`` while ``|flag| <= 1e-3`` and
          ``<a, b> > thres``, zeros otherwise - rate from one probe, thres from the closure's floats (bisection as a
          fallback), the flag tolerance and random on / off cases checked against ``make_gate``."""
        fn = n.output
        rs = np.random.RandomState(0)
        try:
            ok = all(np.array_equal(np.asarray(fn(0.01 * (i + 1), x)), x)
                     for i, x in enumerate(rs.randn(3, n.size_in)))
        except Exception:
            ok = False
        if ok and n.size_in == n.size_out:
            return ("identity",)
        for recognise in (_recognise_cleanup, _recognise_gate):
            try:
                native = recognise(fn, n.size_in, n.size_out)
            except Exception:                    # noqa: BLE001 - a function that does not behave like the pattern
                native = None
            if native is not None:
                return native
        raise fe.BuildError(
            f"{n!r}: Python function nodes with inputs cannot run inside the device step loop; "
            "tag the node with node.native = ('identity',) | ('cleanup', table) | ('gate', d, thres, rate)")

    # -- ensembles -------------------------------------------------------------------------
    def _build_ensemble_params(self, e):
        rng = np.random.RandomState(self.seed_of[id(e)])
        n, d = e.n_neurons, e.dimensions
        nd = _neuron_desc(e.neuron_type)
        enc = _sample(e.encoders, n, d, rng, fe.ScatteredHypersphere(surface=True))
        if enc.shape != (n, d):
            raise fe.BuildError(f"{e!r}: encoders must be ({n}, {d}), got {enc.shape}")
        if getattr(e, "normalize_encoders", True):
            enc = enc / np.linalg.norm(enc, axis=1, keepdims=True)
        max_rates = _sample(e.max_rates, n, None, rng, fe.Uniform(200, 400)).reshape(-1)
        intercepts = _sample(e.intercepts, n, None, rng, fe.Uniform(-1.0, 0.9)).reshape(-1)
        if max_rates.size != n or intercepts.size != n:
            raise fe.BuildError(f"{e!r}: max_rates / intercepts must have {n} entries")
        gain_given = getattr(e, "gain", fe.Default)
        if gain_given is not fe.Default and gain_given is not None and type(gain_given).__name__ != "DefaultType":
            gain = np.asarray(gain_given, dtype=float)
            bias = np.asarray(e.bias, dtype=float)
        else:
            gain, bias = e.neuron_type.gain_bias(max_rates, intercepts)
        scaled = enc * (gain / e.radius)[:, None]
        self.built_ens[id(e)] = be = BuiltEnsemble(
            encoders=enc, scaled_encoders=scaled, gain=gain, bias=bias, max_rates=max_rates,
            intercepts=intercepts, eval_points=None, neuron=nd, radius=float(e.radius),
            seed=self.seed_of[id(e)], decoder_cache={})
        self.model.params[e] = be

    def _eval_points(self, e, conn_eval_points=None):
        be = self.built_ens[id(e)]
        if conn_eval_points is not None:
            return np.asarray(conn_eval_points, dtype=float)
        if be.eval_points is None:
            rng = np.random.RandomState(be.seed ^ 0x5EED)
            given = getattr(e, "eval_points", fe.Default)
            if given is not fe.Default and type(given).__name__ != "DefaultType" and not hasattr(given, "sample"):
                be.eval_points = np.asarray(given, dtype=float)
            else:
                m = getattr(e, "n_eval_points", fe.Default)
                if m is fe.Default or m is None or type(m).__name__ == "DefaultType":
                    m = self.n_eval_points
                if m is None:
                    m = max(int(np.clip(500 * e.dimensions, 750, 2500)), 2 * e.n_neurons)
                dist = given if hasattr(given, "sample") else fe.ScatteredHypersphere(surface=False)
                be.eval_points = np.asarray(dist.sample(int(m), e.dimensions, rng=rng)) * e.radius
        return be.eval_points

    def _request_decoders(self, conn, e, pre_start, pre_len, T, callback):
        """Queue ``callback(W)`` with ``W = T @ decoders(conn.function)`` (size_out, n): all decoded
        connections of one ensemble are solved together from a single factorisation."""
        self.pending.setdefault(id(e), []).append((conn, pre_start, pre_len, T, callback))

    def _targets(self, conn, X, pre_start, pre_len):
        fn = conn.function
        Xs = X[:, pre_start:pre_start + pre_len]
        if fn is None:
            return Xs
        if hasattr(fn, "batch"):
            Y = np.asarray(fn.batch(Xs), dtype=float)
        else:
            Y = self._vectorised_targets(fn, Xs)
            if Y is None:
                Y = np.stack([np.asarray(fn(x), dtype=float).reshape(-1) for x in Xs])
        return Y.reshape(X.shape[0], -1)

    @staticmethod
    def _vectorised_targets(fn, Xs):
        """``fn`` applied to all eval points in ONE call when it is row-wise (``np.square`` of the product ensembles, reference
        ``binding.py:316-317``: 8 128 ensembles x 1 500 eval points were 12 M Python calls, 22 s of a config-3 build): accepted
        only if the batched result has one row per point and equals the per-point calls bit for bit on the first, second
        and last point and on sixteen more drawn with a fixed seed; otherwise None (the caller loops)."""
        m = Xs.shape[0]
        if m < 4:
            return None
        try:
            Y = np.asarray(fn(Xs), dtype=float)
        except Exception:                    # noqa: BLE001 - a function written for one point: loop
            return None
        if Y.ndim == 1 and Y.shape[0] == m:
            Y = Y[:, None]
        if Y.ndim != 2 or Y.shape[0] != m:
            return None
        # (a function that broadcasts on the first row - np.where(x[0] > 0, x, -x) - agrees with the per-point calls on row 0 by
        #  construction and on any other single row with probability ~1/2: sixteen rows drawn with a fixed seed make an
        #  accidental pass a 2^-16 event, at sixteen calls against the millions the batched call saves)
        pick = sorted({0, 1, m - 1} | set(np.random.RandomState(0x5EED).choice(m, size=min(16, m), replace=False).tolist()))
        for i in pick:
            yi = np.asarray(fn(Xs[i]), dtype=float).reshape(-1)
            if yi.shape != Y[i].shape or not np.array_equal(yi, Y[i], equal_nan=True):
                return None
        return Y

    def _solve_pending(self, e):
        items = self.pending.pop(id(e), [])
        if not items:
            return
        be = self.built_ens[id(e)]
        groups = {}
        for it in items:
            conn = it[0]
            solver = conn.solver
            sname = type(solver).__name__
            if getattr(solver, "weights", False):
                raise fe.BuildError("weight solvers are not supported")
            if sname == "NoSolver":
                vals = getattr(solver, "values", None)
                D = np.zeros((conn.size_mid, e.n_neurons)) if vals is None else np.asarray(vals, dtype=float).T
                self._finish_decode(it, D)
                continue
            if sname != "LstsqL2":
                raise fe.BuildError(f"unsupported solver {sname} (LstsqL2, NoSolver are available)")
            key = (float(solver.reg), id(conn.eval_points) if conn.eval_points is not None else None)
            groups.setdefault(key, []).append(it)
        for (reg, _), lst in groups.items():
            X = self._eval_points(e, lst[0][0].eval_points)
            Ys = [self._targets(it[0], X, it[1], it[2]) for it in lst]
            widths = [y.shape[1] for y in Ys]
            Y = np.concatenate(Ys, axis=1)
            nz = np.any(Y != 0, axis=0)
            D = np.zeros((Y.shape[1], e.n_neurons))
            if nz.any():
                D[nz] = solve_decoders(X, be.scaled_encoders, be.bias, be.neuron, Y[:, nz], reg=reg,
                                       backend=self.solver_backend).T
            off = 0
            for it, w in zip(lst, widths):
                self._finish_decode(it, D[off:off + w])
                off += w

    @staticmethod
    def _finish_decode(item, D):
        _, _, _, T, callback = item
        T = np.asarray(T, dtype=float)
        callback(D * float(T) if T.ndim == 0 else T @ D)

    def _neuron_range(self, e):
        """[lo, hi): the neurons of dense ensemble ``e`` this rank steps (all of them without neuron sharding)."""
        if self.neuron_shard is None or id(e) in self.replicated or id(e) in getattr(self, "block_of", {}):
            return 0, e.n_neurons
        rank, world = self.neuron_shard
        per = -(-e.n_neurons // world)
        return min(e.n_neurons, rank * per), min(e.n_neurons, (rank + 1) * per)

    def _local_count(self, e):
        """Neurons of ``e`` in this rank's arrays: every rank's share is padded to the same size (silent neurons: zero
        bias, encoders and decoders), so that the signal layout - and with it the exchange ranges - is the same on all ranks."""
        if self.neuron_shard is None or id(e) in self.replicated or id(e) in getattr(self, "block_of", {}):
            return e.n_neurons
        return -(-e.n_neurons // self.neuron_shard[1])

    @staticmethod
    def _pad_rows(a, n):
        a = np.asarray(a, dtype=float)
        if a.shape[0] == n:
            return a
        out = np.zeros((n,) + a.shape[1:])
        out[:a.shape[0]] = a
        return out

    def _is_sharded(self, e):
        return self.neuron_shard is not None and id(e) not in self.replicated

    def _is_local(self, e):
        """False for EnsembleArray members owned by another rank of a VCO-sharded build."""
        if id(e) not in self.block_of:
            return True
        blk, i = self.block_of[id(e)]
        return blk["range"][0] <= i < blk["range"][1]

    # -- connections -----------------------------------------------------------------------
    def _target_ref(self, post):
        obj, start, length = _contig(post)
        k = _kind(obj)
        if k == "node":
            if id(obj) not in self.node_in:
                raise fe.BuildError(f"{obj!r} has no input")
            full = self.node_in[id(obj)]
        elif k == "ensemble":
            full = self.ens_in[id(obj)]
        elif k == "neurons":
            ens = obj.ensemble
            if id(ens) in self.block_of:
                raise fe.BuildError("direct neuron input into an EnsembleArray member is not supported")
            if self._is_sharded(ens) and length is not None:
                raise fe.BuildError("slices of the neurons of a neuron-sharded ensemble are not supported")
            full = self.ens_J[id(ens)]
        elif k == "rule":
            full = self._rule_input(obj)
        else:
            raise fe.BuildError(f"cannot connect into {obj!r}")
        return full if length is None else full.slice(start, length)

    def _rule_input(self, rule):
        if id(rule) not in self.rule_in:
            is_voja = type(rule.learning_rule_type).__name__ == "Voja"
            # Voja's learning signal is Reset to 1 each step, PES's error to 0 (Appendix A.7/A.8):
            # kept in the R arena; Voja's "+1" is applied where the rule reads it.
            self.rule_in[id(rule)] = self.alloc("R", 1 if is_voja else rule.size_in)
        return self.rule_in[id(rule)]

    def _apply_synapse(self, conn, src, dst, size, gain=1.0):
        """src (already transformed, ``size`` wide) -> [Lowpass] -> dst accumulation."""
        syn = conn.synapse
        if syn is None:
            self.op("axpy", dst=dst, src=src, len=size, alpha=gain, mode="inc")
            return
        tau = float(syn.tau)
        state = self.alloc("S", size)
        self.op("lowpass", dst=state, src=src, len=size, a=_lowpass_coeff(tau, self.dt), gain=gain)
        self.op("axpy", dst=dst, src=state, len=size, alpha=1.0, mode="inc")

    def _lower_connection(self, c):
        pre_obj, pre_start, pre_len = _contig(c.pre)
        pk = _kind(pre_obj)
        dst = self._target_ref(c.post)
        size_out = dst.len
        T = np.asarray(c.transform, dtype=float)
        rule = getattr(c, "learning_rule_type", None)
        bc = BuiltConnection(weights=None, learned_buffer=None)
        self.model.params[c] = bc

        if pk == "node":
            full = self.node_out[id(pre_obj)]
            src = full if pre_len is None else full.slice(pre_start, pre_len)
            if c.function is not None:
                raise fe.BuildError("functions on connections from Nodes are not supported")
            post_obj = _contig(c.post)[0]
            if rule is not None:
                if type(rule).__name__ != "Voja" or _kind(post_obj) != "ensemble" or T.ndim != 0:
                    raise fe.BuildError("only Voja on a plain Node -> Ensemble connection is supported")
                self._lower_voja(c, src, post_obj, float(T), rule)
                return
            if _kind(post_obj) == "neurons" and T.ndim == 2:
                lo, hi = self._neuron_range(post_obj.ensemble)      # direct neuron input: this rank's rows of the transform
                T = self._pad_rows(T[lo:hi], self._local_count(post_obj.ensemble))
            if T.ndim == 0:
                self._apply_synapse(c, src, dst, size_out, gain=float(T))
            elif T.ndim == 1:
                raise fe.BuildError("diagonal transforms are not supported; pass a full matrix")
            else:
                self._lower_matrix(c, T, src, dst)
            bc.weights = T
            return

        if pk == "neurons":
            ens = pre_obj.ensemble
            if id(ens) in self.block_of:
                raise fe.BuildError("connections from the neurons of an EnsembleArray member are not supported")
            if self._is_sharded(ens):
                raise fe.BuildError("connections from the neurons of a neuron-sharded ensemble are not supported")
            full = self.ens_spk[id(ens)]
            src = full if pre_len is None else full.slice(pre_start, pre_len)
            if T.ndim == 0:
                self._apply_synapse(c, src, dst, size_out, gain=float(T))
            else:
                self._lower_matrix(c, T, src, dst)
            return

        if pk != "ensemble":
            raise fe.BuildError(f"cannot connect from {pre_obj!r}")
        e = pre_obj
        post_obj = _contig(c.post)[0]
        if _kind(post_obj) == "neurons" and self._is_sharded(post_obj.ensemble) and T.ndim == 2:
            # (Node -> neurons slices the rows of its transform above; a decoded connection would need its decoders solved for
            #  this rank's rows only - the reference's inhibition, associativememory.py:47-49, comes from a Node)
            raise fe.BuildError("a decoded connection with a matrix transform into the neurons of a neuron-sharded ensemble is "
                                "not supported: replicate the post ensemble (build(..., replicate=[...])) or drive it from a Node")
        if pre_len is None:
            pre_start, pre_len = 0, e.dimensions
        if rule is not None and type(rule).__name__ == "PES":
            self._lower_pes(c, e, pre_start, pre_len, T, dst, rule)
            return
        if rule is not None:
            raise fe.BuildError(f"unsupported learning rule {rule!r} on a decoded connection")
        w = self.alloc("W", size_out)
        if self._is_local(e):               # other ranks' VCOs: same signal layout, no decoders
            def done(W, e=e, w=w, bc=bc):
                bc.weights = W
                self._register_rows(e, W, w)
            self._request_decoders(c, e, pre_start, pre_len, T, done)
        elif self.neuron_shard is not None:
            # neuron sharding: the slot is this rank's (zero) share of a sum that the exchange completes in place - it has to
            # be cleared again every timestep
            self.op("fill", dst=w, len=size_out, value=0.0, partial_zero=True)
        self._apply_synapse(c, w, dst, size_out)

    def _lower_matrix(self, c, T, src, dst):
        rows, cols = T.shape
        if cols != src.len or rows != dst.len:
            raise fe.BuildError(f"transform {T.shape} does not map {src.len} -> {dst.len}")
        b = self.model.add_buffer(np.ascontiguousarray(T), f"transform_{len(self.model.buffers)}")
        dft = dft_structure(T)          # the matrix is kept (oracle, f64 mode); the f32 device path may use an FFT
        if c.synapse is None:
            self.op("matvec", dst=dst, src=src, rows=rows, cols=cols, w=b, mode="inc", dft=dft)
        else:
            w = self.alloc("W", rows)
            self.op("matvec", dst=w, src=src, rows=rows, cols=cols, w=b, mode="set", dft=dft)
            self._apply_synapse(c, w, dst, rows)

    def _register_rows(self, e, W, w_ref):
        """Decoded rows of ensemble ``e``: row r of W (n,) lands in sig[w_ref + r]."""
        if id(e) in self.block_of:
            blk, i = self.block_of[id(e)]
            for r in range(W.shape[0]):
                if np.any(W[r]):
                    blk["rows"][i].append((W[r], w_ref.slice(r, 1)))
        else:
            lo, hi = self._neuron_range(e)
            self.dense_rows[id(e)].append((self._pad_rows(W[:, lo:hi].T, self._local_count(e)).T, w_ref))

    def _lower_pes(self, c, e, pre_start, pre_len, T, dst, rule):
        if id(e) in self.block_of:
            raise fe.BuildError("PES on an EnsembleArray member is not supported")
        size_out, n = dst.len, e.n_neurons
        lo, hi = self._neuron_range(e)           # neuron sharding: this rank learns its columns of the decoder matrix
        nl = self._local_count(e)
        b = self.model.add_buffer(np.zeros((size_out, nl)), f"pes_decoders_{len(self.model.buffers)}", role="learned")
        self.model.params[c].learned_buffer = b
        self.model.params[c].learned_columns = (lo, hi)

        def done(W, b=b, c=c, lo=lo, hi=hi):
            self.model.buffers[b][:, :hi - lo] = W[:, lo:hi]
            self.model.params[c].weights = W
        self._request_decoders(c, e, pre_start, pre_len, T, done)
        w = self.alloc("W", size_out)
        spk = self.ens_spk[id(e)]
        self.op("matvec", dst=w, src=spk, rows=size_out, cols=nl, w=b, mode="set", partial_out=self._is_sharded(e))
        self._apply_synapse(c, w, dst, size_out)
        # filtered pre activities (pre_synapse) and the fused delta + increment (Appendix A.7); kappa uses the size of the
        # WHOLE pre population (nengo: -lr * dt / n_neurons)
        act = self.alloc("S", nl)
        tau = float(rule.pre_synapse.tau) if rule.pre_synapse is not None else 0.0
        err = self._rule_input(c.learning_rule)
        self.op("pes", w=b, rows=size_out, cols=nl, err=err, act=act,
                kappa=-float(rule.learning_rate) * self.dt / n)
        self.op("lowpass", dst=act, src=spk, len=nl, a=_lowpass_coeff(tau, self.dt) if tau > 0 else 0.0, gain=1.0)
        self.learned[id(c)] = b

    def _lower_voja(self, c, src, post_ens, alpha, rule):
        """Node -> Ensemble with Voja: the key reaches the ensemble through its encoders as usual;
        the rule then moves the (scaled) encoders of spiking neurons towards the key (A.8)."""
        if id(post_ens) in self.block_of:
            raise fe.BuildError("Voja on an EnsembleArray member is not supported")
        d = post_ens.dimensions
        dst = self.ens_in[id(post_ens)]
        if c.synapse is not None:
            raise fe.BuildError("Voja connections with a synapse are not supported (reference uses synapse=None)")
        key = self.alloc("W", d)   # the connection's weighted output = pre_decoded of the rule
        self.op("axpy", dst=key, src=src, len=d, alpha=alpha, mode="set")
        self.op("axpy", dst=dst, src=key, len=d, alpha=1.0, mode="inc")
        if rule.post_synapse is not None:
            raise fe.BuildError("Voja with a post_synapse is not supported (reference uses post_synapse=None)")
        be = self.built_ens[id(post_ens)]
        be.voja = dict(key=key, learn=self._rule_input(c.learning_rule), lr=float(rule.learning_rate))
        self.model.params[c].weights = alpha

    # -- emit ensembles --------------------------------------------------------------------
    def _emit_dense_ensemble(self, e):
        be = self.built_ens[id(e)]
        d = e.dimensions
        lo, hi = self._neuron_range(e)
        n = self._local_count(e)
        part = self._is_sharded(e)
        J, spk, x = self.ens_J[id(e)], self.ens_spk[id(e)], self.ens_in[id(e)]
        bias = self.alloc("C", n, init=self._pad_rows(be.bias[lo:hi], n))
        role = "learned" if hasattr(be, "voja") else "param"
        eb = self.model.add_buffer(np.ascontiguousarray(self._pad_rows(be.scaled_encoders[lo:hi], n)), f"encoders_{e.label}", role=role)
        be.encoder_buffer = eb
        be.neuron_range = (lo, hi)
        self.op("axpy", dst=J, src=bias, len=n, alpha=1.0, mode="inc")
        self.op("matvec", dst=J, src=x, rows=n, cols=d, w=eb, mode="inc", local_out=part)
        vb = self.model.add_buffer(np.zeros(n), f"voltage_{e.label}", role="state")
        rb = self.model.add_buffer(np.zeros(n), f"refractory_{e.label}", role="state")
        be.state_buffers = (vb, rb)
        self.op("neurons", j=J, out=spk, n=n, v=vb, r=rb, neuron=be.neuron,
                amp=be.neuron["amplitude"] / self.dt if be.neuron["type"] == "lif" else be.neuron["amplitude"])
        for W, w_ref in self.dense_rows[id(e)]:
            b = self.model.add_buffer(np.ascontiguousarray(W), f"decoders_{e.label}_{len(self.model.buffers)}")
            self.op("matvec", dst=w_ref, src=spk, rows=W.shape[0], cols=n, w=b, mode="set", partial_out=part)
        if hasattr(be, "voja"):
            v = be.voja
            self.op("voja", w=eb, rows=n, cols=d, spk=spk, key=v["key"], learn=v["learn"],
                    lr_dt=v["lr"] * self.dt,
                    scale_buf=self.model.add_buffer(self._pad_rows((be.gain / be.radius)[lo:hi], n), f"voja_scale_{e.label}"))

    def _live_elements(self, probes):
        """Which signal elements can influence anything observable (a probe, a neuron, a learning rule)?

        A decoded row of an EnsembleArray member costs registers and arithmetic on every neuron-step, and the reference's
        graphs contain rows nothing ever looks at: ``PathIntegration`` decodes all three dimensions of every oscillator
        into ``oscillators.output`` (``pathintegration.py:162-167``) but the read-out ``to_SSP`` has an all-zero column
        for each frequency dimension (``get_from_Fourier``, ``:824-844``).  Backward data-flow over the lowered
        operators: an element is live if a probe samples it, if it is an ensemble / neuron / rule / function-node input,
        or if a live element is computed from it - through a constant matrix only where the column is non-zero.
        Returns {arena: bool array}; ``_emit_block`` drops the rows whose destination is dead (their value is then never
        produced; nothing observable changes)."""
        live = {a: np.zeros(self.arena_size[a], dtype=bool) for a in "RWSCT"}

        def mark(r):
            live[r.arena][r.off:r.off + r.len] = True

        for p in self.model.probes:
            if "src" in p:
                unused = getattr(p["probe"], "unused", None)
                if unused is None:
                    mark(p["src"])
                else:
                    # a probe that declares elements nobody will look at (the shard probe of the sharded path integrator: the
                    # frequency rows of the oscillator outputs meet all-zero columns of to_SSP) does not keep them alive:
                    # they are sampled as the zeros they are initialised with
                    r = p["src"]
                    keep = ~np.asarray(unused, dtype=bool).reshape(-1)
                    if keep.size != r.len:
                        raise fe.BuildError(f"probe.unused has {keep.size} entries for a probe of width {r.len}")
                    live[r.arena][r.off:r.off + r.len] |= keep
        for table in (self.ens_in, self.ens_J, self.rule_in):
            for r in table.values():
                mark(r)
        for blk in self.blocks:
            mark(blk["x"])
        for o in self.raw_ops:
            if o["kind"] == "cleanup":
                mark(Ref(o["src"].arena, o["src"].off, o["cols"]))
            elif o["kind"] == "gate":
                mark(Ref(o["src"].arena, o["src"].off, 2 * o["d"] + 1))
            elif o["kind"] == "voja":
                mark(Ref(o["key"].arena, o["key"].off, o["cols"]))
                mark(Ref(o["spk"].arena, o["spk"].off, o["rows"]))
            elif o["kind"] == "pes":
                mark(Ref(o["act"].arena, o["act"].off, o["cols"]))
                mark(Ref(o["err"].arena, o["err"].off, o["rows"]))
        changed = True
        while changed:
            changed = False
            for o in reversed(self.raw_ops):
                k = o["kind"]
                if k in ("axpy", "lowpass"):
                    d, sr, n = o["dst"], o["src"], o["len"]
                    need = live[d.arena][d.off:d.off + n]
                    cur = live[sr.arena][sr.off:sr.off + n]
                    if np.any(need & ~cur):
                        cur |= need
                        changed = True
                elif k == "matvec":
                    d, sr = o["dst"], o["src"]
                    rows_live = live[d.arena][d.off:d.off + o["rows"]]
                    if not rows_live.any():
                        continue
                    if self.model.buffer_meta[o["w"]]["role"] == "param":
                        need = np.any(self.model.buffers[o["w"]][rows_live] != 0, axis=0)
                    else:
                        need = np.ones(o["cols"], dtype=bool)
                    cur = live[sr.arena][sr.off:sr.off + o["cols"]]
                    if np.any(need & ~cur):
                        cur |= need
                        changed = True
        return live

    def _row_is_live(self, ref):
        return bool(self._live[ref.arena][ref.off:ref.off + ref.len].any())

    def _emit_block(self, blk):
        ens = blk["ens"]
        K_all = len(ens)
        lo, hi = blk["range"]
        e0 = ens[0]
        n, din = e0.n_neurons, e0.dimensions
        K = hi - lo
        if K == 0:
            return
        n_rows = sum(len(blk["rows"][i]) for i in range(lo, hi))
        for i in range(lo, hi):
            blk["rows"][i] = [rw for rw in blk["rows"][i] if self._row_is_live(rw[1])]
        self.model.stats["dead_decoded_rows"] = self.model.stats.get("dead_decoded_rows", 0) + \
            n_rows - sum(len(blk["rows"][i]) for i in range(lo, hi))
        dout = max(1, max(len(blk["rows"][i]) for i in range(lo, hi)))
        enc = np.zeros((K, din, n))
        bias = np.zeros((K, n))
        dec = np.zeros((K, dout, n))
        trash = self.alloc("W", 1)
        dst_idx = np.empty((K, dout), dtype=object)
        nd = self.built_ens[id(e0)].neuron
        amp = nd["amplitude"] / self.dt if nd["type"] == "lif" else nd["amplitude"]
        for i in range(lo, hi):
            be = self.built_ens[id(ens[i])]
            enc[i - lo] = be.scaled_encoders.T
            bias[i - lo] = be.bias
            rows = blk["rows"][i]
            for r in range(dout):
                if r < len(rows):
                    dec[i - lo, r] = rows[r][0] * amp
                    dst_idx[i - lo, r] = rows[r][1]
                else:
                    dst_idx[i - lo, r] = trash
        m = self.model
        label = getattr(blk["net"], "label", None) or "ensarray"
        x = blk["x"].slice(lo * din, K * din)
        self.op("ensarray", x=x, K=K, n=n, din=din, dout=dout,
                enc=m.add_buffer(enc, f"{label}_enc"), bias=m.add_buffer(bias, f"{label}_bias"),
                dec=m.add_buffer(dec, f"{label}_dec"), dst_refs=dst_idx,
                v=m.add_buffer(np.zeros((K, n)), f"{label}_voltage", role="state"),
                r=m.add_buffer(np.zeros((K, n)), f"{label}_refractory", role="state"),
                neuron=nd, label=label, k_lo=lo, k_total=K_all, partial_out=bool(self.neuron_shard is not None and K < K_all))

    # -- probes ----------------------------------------------------------------------------
    def _lower_probe(self, p):
        obj, start, length = _contig(p.target)
        k = _kind(obj)
        every = 1 if p.sample_every is None else max(1, int(round(p.sample_every / self.dt)))
        attr = p.attr
        if k == "connection" and attr == "weights":
            bc = self.model.params.get(obj)
            if bc is None or bc.learned_buffer is None:
                raise fe.BuildError("Probe(conn, 'weights') needs a connection with a learning rule")
            buf = self.model.buffers[bc.learned_buffer]
            self.model.probes.append({"probe": p, "buf": bc.learned_buffer, "shape": buf.shape, "every": every})
            return
        if k == "rule" and attr == "scaled_encoders":
            post = _contig(obj.connection.post)[0]
            be = self.built_ens[id(post)]
            self.model.probes.append({"probe": p, "buf": ("encoders_of", id(post)),
                                      "shape": be.scaled_encoders.shape, "every": every, "ens": post})
            return
        if k == "node":
            full = self.node_out[id(obj)]
        elif k == "ensemble":
            if attr not in ("decoded_output", "output"):
                raise fe.BuildError(f"unsupported ensemble probe attr {attr!r}")
            e = obj
            fake = type("C", (), dict(function=None, solver=fe.LstsqL2(), eval_points=None, size_mid=e.dimensions))()
            full = self.alloc("W", e.dimensions)
            if self._is_local(e):
                def done(W, e=e, full=full, p=p):
                    self.model.params[p] = BuiltConnection(weights=W, learned_buffer=None)    # identity decoders of the probe
                    self._register_rows(e, W, full)
                self._request_decoders(fake, e, 0, e.dimensions, 1.0, done)
        elif k == "neurons":
            e = obj.ensemble
            if id(e) in self.block_of:
                raise fe.BuildError("probing neurons of an EnsembleArray member is not supported yet")
            if self._is_sharded(e):
                raise fe.BuildError("probing the neurons of a neuron-sharded ensemble is not supported")
            full = self.ens_spk[id(e)]
        else:
            raise fe.BuildError(f"unsupported probe target {obj!r}")
        src = full if length is None else full.slice(start, length)
        if p.synapse is not None:
            state = self.alloc("S", src.len)
            self.op("lowpass", dst=state, src=src, len=src.len, a=_lowpass_coeff(float(p.synapse.tau), self.dt), gain=1.0)
            src = state
        self.model.probes.append({"probe": p, "src": src, "width": src.len, "every": every})

    # -- finalise: resolve offsets, merge, schedule ------------------------------------------
    def _finalise(self):
        m = self.model
        base, off = {}, 0
        for a in "RWSCT":
            base[a] = off
            off += self.arena_size[a]
        m.sig_size, m.arena_base, m.arena_size = off, base, dict(self.arena_size)
        m.sig_init = np.zeros(off)
        for r, vals in self.inits:
            m.sig_init[base[r.arena] + r.off: base[r.arena] + r.off + r.len] = vals

        def A(r):
            return base[r.arena] + r.off

        for table in m.tables:
            table["dst"] = A(table.pop("ref"))
        ops = []
        for o in self.raw_ops:
            o = dict(o)
            for key in ("dst", "src", "x", "j", "out", "err", "act", "spk", "key", "learn"):
                if isinstance(o.get(key), Ref):
                    o[key] = A(o[key])
            if o["kind"] == "ensarray":
                refs = o.pop("dst_refs")
                idx = np.array([[A(r) for r in row] for row in refs], dtype=np.int32)
                o["dst_idx"] = m.add_buffer(idx, f"{o['label']}_dst_idx", role="index")
            ops.append(o)
        for p in m.probes:
            if "src" in p:
                p["src"] = A(p["src"])
        for name, table in (("ens_in", self.ens_in), ("node_in", self.node_in), ("node_out", self.node_out),
                            ("ens_spk", self.ens_spk)):
            for key, r in table.items():
                m.sig[(name, key)] = (A(r), r.len)
        ops = merge_ops(ops, m)
        if self.prune:
            ops = prune_ops(ops, m)
        r_allocs = [(base["R"] + off, ln) for off, ln in self.r_allocs]
        collapse = None
        if self.collapse:
            def collapse(sub, protected):
                sub, st = collapse_glue(sub, m, op_access, protected)
                m.stats.update({"glue_" + k: v for k, v in st.items()})
                for o in sub:
                    if o["kind"] == "lincomb":       # the term lists travel as buffers (signal offsets, coefficients)
                        o["srcs_buf"] = m.add_buffer(np.asarray(o["srcs"], dtype=np.int32), f"lincomb_src_{len(m.buffers)}", role="index")
                        o["alphas_buf"] = m.add_buffer(np.asarray(o["alphas"], dtype=np.float64), f"lincomb_alpha_{len(m.buffers)}")
                return sub
        m.ops = stage_ops(ops, m, r_allocs, schedule_ops, op_access, _overlap, enable=self.staged, collapse=collapse)
        if self.neuron_shard is not None:
            shard_phases(m, self.neuron_shard)
        m.stats.update(n_raw_ops=len(self.raw_ops), n_ops=len(m.ops), sig_size=m.sig_size,
                       n_buffers=len(m.buffers))


# --------------------------------------------------------------------------------------------
# operator merging (nengo's optimizer pass, Appendix A.12)
# --------------------------------------------------------------------------------------------
def merge_ops(ops, model):
    out = []
    groups = {}
    for o in ops:
        k = o["kind"]
        if k == "fill":
            key = ("fill", o["value"], bool(o.get("partial_zero")))
        elif k == "axpy":
            key = ("axpy", o["alpha"], o["mode"], o["dst"] - o["src"])
        elif k == "lowpass":
            key = ("lowpass", o["a"], o["gain"], o["dst"] - o["src"])
        elif k == "matvec" and o["rows"] * o["cols"] <= 64 and model.buffer_meta[o["w"]]["role"] == "param":
            key = ("matvec", o["src"], o["cols"], o["mode"])
        else:
            out.append(o)
            continue
        groups.setdefault(key, []).append(o)
    for key, lst in groups.items():
        lst.sort(key=lambda o: o["dst"])
        cur = None
        for o in lst:
            n = o["rows"] if key[0] == "matvec" else o["len"]
            if cur is not None and o["dst"] == cur["dst"] + (cur["rows"] if key[0] == "matvec" else cur["len"]):
                if key[0] == "matvec":
                    cur["_stack"].append(model.buffers[o["w"]])
                    cur["rows"] += n
                else:
                    cur["len"] += n
                cur["seq"] = min(cur["seq"], o["seq"])
                continue
            if cur is not None:
                out.append(cur)
            cur = dict(o)
            if key[0] == "matvec":
                cur["_stack"] = [model.buffers[o["w"]]]
        if cur is not None:
            out.append(cur)
    for o in out:
        if "_stack" in o:
            st = o.pop("_stack")
            if len(st) > 1:
                o["w"] = model.add_buffer(np.ascontiguousarray(np.vstack(st)), f"stacked_transform_{len(model.buffers)}")
    out.sort(key=lambda o: o["seq"])
    return out


def op_access(o, model):
    """(sets, incs, reads, updates): lists of resources ('s', lo, hi) signal ranges / ('b', id) buffers."""
    k = o["kind"]
    S = lambda off, n: ("s", off, off + n)  # noqa: E731
    B = lambda b: ("b", b)                  # noqa: E731
    if k == "fill":
        return [S(o["dst"], o["len"])], [], [], []
    if k == "table":
        return [S(o["dst"], o["width"])], [], [], []
    if k == "axpy":
        w = [S(o["dst"], o["len"])]
        return (w, [], [S(o["src"], o["len"])], []) if o["mode"] == "set" else ([], w, [S(o["src"], o["len"])], [])
    if k == "lincomb":
        w = [S(o["dst"], o["len"])]
        r = [S(src, o["len"]) for src in o["srcs"]]
        return ([], w, r, []) if o["self"] else (w, [], r, [])
    if k == "matvec":
        w = [S(o["dst"], o["rows"])]
        r = [S(o["src"], o["cols"]), B(o["w"])]
        return (w, [], r, []) if o["mode"] == "set" else ([], w, r, [])
    if k == "lowpass":
        return [], [], [S(o["src"], o["len"])], [S(o["dst"], o["len"])]
    if k == "ensarray":
        idx = np.unique(model.buffers[o["dst_idx"]].reshape(-1))
        runs, start, prev = [], int(idx[0]), int(idx[0])
        for v in idx[1:]:
            v = int(v)
            if v != prev + 1:
                runs.append(S(start, prev - start + 1))
                start = v
            prev = v
        runs.append(S(start, prev - start + 1))
        return runs, [], [S(o["x"], o["K"] * o["din"])], [B(o["v"]), B(o["r"])]
    if k == "neurons":
        return [S(o["out"], o["n"])], [], [S(o["j"], o["n"])], [B(o["v"]), B(o["r"])]
    if k == "pes":
        return [], [], [S(o["err"], o["rows"]), S(o["act"], o["cols"])], [B(o["w"])]
    if k == "voja":
        return [], [], [S(o["spk"], o["rows"]), S(o["key"], o["cols"]), S(o["learn"], 1), B(o["scale_buf"])], [B(o["w"])]
    if k == "cleanup":
        return [S(o["dst"], o["cols"])], [], [S(o["src"], o["cols"]), B(o["w"])] + \
            [B(o[nm]) for nm in ("g_dft", "g_lhs", "g_rhs") if nm in o], []
    if k == "gate":
        return [S(o["dst"], o["d"])], [], [S(o["src"], 2 * o["d"] + 1)], []
    raise fe.BuildError(f"unknown op kind {k}")


def prune_ops(ops, model):
    """Dead-operator elimination: keep only operators whose writes can reach a probe, directly or
    through synapse states / neuron state over later timesteps (fixed point).  Used for the per-rank
    shard of a VCO-sharded model, where the read-out chain runs elsewhere."""
    acc = [op_access(o, model) for o in ops]
    live = []
    for p in model.probes:
        if "src" in p:
            live.append(("s", p["src"], p["src"] + p["width"]))
        else:
            b = p["buf"]
            live.append(("b", b if not isinstance(b, tuple) else -1))
    keep = [False] * len(ops)
    changed = True
    while changed:
        changed = False
        for i, a in enumerate(acc):
            if keep[i]:
                continue
            writes = a[0] + a[1] + a[3]
            if any(_overlap(w, l) for w in writes for l in live):
                keep[i] = True
                live.extend(a[2])
                # an update reads its own previous state; an inc reads the accumulator it adds to
                live.extend(a[1])
                live.extend(a[3])
                changed = True
    return [o for o, k in zip(ops, keep) if k]


def _closure_values(fn, depth=3):
    """Objects reachable through the closure cells (and defaults) of a function, callables followed ``depth`` levels."""
    seen, out, todo = set(), [], [(fn, 0)]
    while todo:
        f, lvl = todo.pop()
        if id(f) in seen:
            continue
        seen.add(id(f))
        cells = []
        for c in (getattr(f, "__closure__", None) or ()):
            try:
                cells.append(c.cell_contents)
            except ValueError:
                pass
        cells.extend(getattr(f, "__defaults__", None) or ())
        for v in cells:
            out.append(v)
            if callable(v) and lvl < depth:
                todo.append((v, lvl + 1))
    return out


def _recognise_cleanup(fn, size_in, size_out):
    if size_in != size_out:
        return None
    tables = [v for v in _closure_values(fn) if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape[1] == size_in and v.shape[0] > 1]
    rs = np.random.RandomState(1)
    for S in tables:
        Sf = np.asarray(S, dtype=float)
        rows = rs.randint(0, Sf.shape[0], size=4)
        probes = [Sf[r] + 0.3 * rs.randn(size_in) / np.sqrt(size_in) for r in rows[:3]] + [rs.randn(size_in)]
        if all(np.array_equal(np.asarray(fn(0.001 * (i + 1), x), dtype=float).reshape(-1), Sf[int(np.argmax(Sf @ x))])
               for i, x in enumerate(probes)):
            from .sspspace import grid_factors_from_table
            return ("cleanup", Sf, grid_factors_from_table(Sf))
    return None


def _recognise_gate(fn, size_in, size_out):
    d = size_out
    if size_in != 2 * d + 1 or d < 1:
        return None
    e0 = np.zeros(d)
    e0[0] = 1.0

    def call(a, b, flag):
        return np.asarray(fn(0.001, np.concatenate([a, b, [flag]])), dtype=float).reshape(-1)

    def on(s, flag=0.0):                     # <s e0, e0> = s exactly
        return bool(np.any(call(s * e0, e0, flag) != 0.0))
    rate = None
    for k in range(1, 12):                   # s - 1 a power of two: the quotient below is exact
        s_hi = 1.0 + 2.0 ** k
        o = call(s_hi * e0, e0, 0.0)
        if o.shape == (d,) and o[0] != 0.0:
            rate = float(o[0] / (s_hi - 1.0))
            if np.any(o[1:] != 0.0):
                return None
            break
    if rate is None:
        return None
    floats = sorted({float(v) for v in _closure_values(fn) if isinstance(v, (int, float, np.floating, np.integer))
                     and not isinstance(v, bool)})
    thres = None
    for th in floats:
        if th < s_hi and not on(th) and on(np.nextafter(th, np.inf)):
            thres = th
            break
    if thres is None:                        # no candidate in the closure: bisect the switching point
        lo, hi = -s_hi, s_hi
        if on(lo) or not on(hi):
            return None
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if mid == lo or mid == hi:
                break
            lo, hi = (lo, mid) if on(mid) else (mid, hi)
        thres = lo
    if not on(s_hi, 1e-3) or on(s_hi, 2e-3) or not on(s_hi, -1e-3) or on(s_hi, -2e-3):
        return None                          # (the flag tolerance of np.allclose(flag, 0, atol=1e-3))
    from .networks.slam import make_gate
    own = make_gate(d, thres, rate)
    rs = np.random.RandomState(2)
    for i in range(6):
        a, b = rs.randn(d), rs.randn(d)
        if i % 2 == 0:
            b = a + 0.1 * rs.randn(d)        # similar vectors: the gate is open when thres is moderate
        x = np.concatenate([a, b, [0.0 if i < 4 else 10.0]])
        if not np.array_equal(np.asarray(fn(0.002, x), dtype=float).reshape(-1), np.asarray(own(0.002, x), dtype=float).reshape(-1)):
            return None
    return ("gate", d, float(thres), float(rate))


def shard_phases(model, shard):
    """Neuron-sharded model (SLAMNetwork on several GPUs): cut the timestep in two around ONE exchange.

    Every rank steps its share of each neuron population, so what its neurons decode is a partial sum.  Linear operators
    commute with the sum; the non-linear consumers of decoded values are neurons, function nodes and - through a synapse,
    i.e. one timestep later - everything else.  So the timestep becomes

        phase 0: every operator except the updates (Lowpass, PES, Voja), on partial data where data is partial
        exchange: all-reduce(sum) of the partial signals the updates (and probes) read - ``model.exchange`` ranges
        phase 1: the updates, then probe sampling

    which is a valid nengo order (updates come last for every signal; they keep their relative order).  A forward pass
    over the operators tracks which elements are replicated / partial and refuses models in which partial data would reach a
    non-linear operator within the timestep (the ensemble in question has to be replicated: ``replicate=``) or in which a
    replicated and a partial term meet in one accumulator."""
    rank, world = shard
    ops = model.ops
    n = model.sig_size
    full = np.zeros(n, dtype=bool)
    part = np.zeros(n, dtype=bool)
    base, size = model.arena_base, model.arena_size
    for a in "SCT":
        full[base[a]:base[a] + size[a]] = True           # states, constants, tables: identical on every rank

    def need_full(lo, ln, what):
        if part[lo:lo + ln].any():
            raise fe.BuildError(f"neuron sharding: {what} would read a partial sum within the timestep - replicate its source "
                                "ensemble (build(..., replicate=[...]))")

    exchange = []
    phase = []
    for o in ops:
        k = o["kind"]
        ph = 1 if k in ("lowpass", "pes", "voja") else 0
        phase.append(ph)
        if k == "fill":
            sl = slice(o["dst"], o["dst"] + o["len"])
            full[sl] = o["value"] != 0.0
            part[sl] = bool(o.get("partial_zero"))       # another rank's share of a decoded sum: this rank adds 0
        elif k == "table":
            sl = slice(o["dst"], o["dst"] + o["width"])
            full[sl], part[sl] = True, False
        elif k == "axpy":
            d, sr = slice(o["dst"], o["dst"] + o["len"]), slice(o["src"], o["src"] + o["len"])
            if o["mode"] == "set":
                full[d], part[d] = full[sr].copy(), part[sr].copy()
            else:
                full[d] |= full[sr]
                part[d] |= part[sr]
        elif k == "lincomb":         # folded glue: linear, so partial where any of its sources is
            d = slice(o["dst"], o["dst"] + o["len"])
            f_new = np.full(o["len"], o["const"] != 0.0)
            p_new = np.zeros(o["len"], bool)
            for src in o["srcs"]:
                f_new |= full[src:src + o["len"]]
                p_new |= part[src:src + o["len"]]
            if o["self"]:
                full[d] |= f_new
                part[d] |= p_new
            else:
                full[d], part[d] = f_new, p_new
        elif k == "matvec":
            d = slice(o["dst"], o["dst"] + o["rows"])
            sr = slice(o["src"], o["src"] + o["cols"])
            if o.get("partial_out"):
                f_new, p_new = np.zeros(o["rows"], bool), np.ones(o["rows"], bool)
            elif o.get("local_out"):
                need_full(o["src"], o["cols"], "an encoder product")
                f_new, p_new = np.ones(o["rows"], bool), np.zeros(o["rows"], bool)
            else:
                if model.buffer_meta[o["w"]]["role"] == "param":
                    nz = model.buffers[o["w"]] != 0
                    f_new, p_new = (nz & full[sr][None, :]).any(axis=1), (nz & part[sr][None, :]).any(axis=1)
                else:
                    f_new, p_new = np.full(o["rows"], full[sr].any()), np.full(o["rows"], part[sr].any())
            if o["mode"] == "set":
                full[d], part[d] = f_new, p_new
            else:
                full[d] |= f_new
                part[d] |= p_new
        elif k == "ensarray":
            need_full(o["x"], o["K"] * o["din"], "an ensemble array")
            idx = model.buffers[o["dst_idx"]].reshape(-1)
            full[idx], part[idx] = (not o.get("partial_out")), bool(o.get("partial_out"))
        elif k == "neurons":
            need_full(o["j"], o["n"], "a neuron population")      # (a partial current must never reach the non-linearity)
            sl = slice(o["out"], o["out"] + o["n"])
            full[sl], part[sl] = True, False             # (a rank's own spike vector: local, never exchanged)
        elif k in ("gate", "cleanup"):
            ln = 2 * o["d"] + 1 if k == "gate" else o["cols"]
            need_full(o["src"], ln, f"the {k} node")
            w = o["d"] if k == "gate" else o["cols"]
            full[o["dst"]:o["dst"] + w], part[o["dst"]:o["dst"] + w] = True, False
        elif k == "lowpass":
            sr = slice(o["src"], o["src"] + o["len"])
            if part[sr].any():
                exchange.append((o["src"], o["src"] + o["len"]))
        elif k == "pes":
            need_full(o["err"], o["rows"], "the PES error")
        elif k == "voja":
            need_full(o["key"], o["cols"], "the Voja key")
            need_full(o["learn"], 1, "the Voja learning signal")
    for p in model.probes:
        if "src" in p and part[p["src"]:p["src"] + p["width"]].any():
            exchange.append((p["src"], p["src"] + p["width"]))
    # only the partial elements of what the updates read are summed (a merged Lowpass may span a replicated vector too)
    want = np.zeros(n, dtype=bool)
    for lo, hi in exchange:
        want[lo:hi] = True
    want &= part
    if (want & full).any():
        bad = int(np.flatnonzero(want & full)[0])
        raise fe.BuildError(f"neuron sharding: a replicated and a partial term are added into one signal (element {bad}): "
                            "the all-reduce would count the replicated one once per rank")
    edges = np.flatnonzero(np.diff(np.concatenate(([0], want.view(np.int8), [0]))))
    merged = []
    for lo, hi in zip(edges[::2], edges[1::2]):
        # short gaps of elements nobody writes (dropped dead rows: zero on every rank) are summed along: one range per
        # ensemble array instead of one per ensemble
        if merged and lo - merged[-1][1] <= 8 and not (full[merged[-1][1]:lo] | part[merged[-1][1]:lo]).any():
            merged[-1][1] = int(hi)
        else:
            merged.append([int(lo), int(hi)])
    merged = [(lo, hi) for lo, hi in merged]
    a_ops = [dict(o) for o, ph in zip(ops, phase) if ph == 0]
    b_ops = [dict(o) for o, ph in zip(ops, phase) if ph == 1]
    out = []
    level0 = 0
    for ph, sub in ((0, a_ops), (1, b_ops)):
        for o in sub:
            for key in ("level", "micro"):
                o.pop(key, None)
        sched = schedule_ops(sub, model)
        for o in sched:
            o["level"] += level0
            o["phase"] = ph
            o["stage"], o["border"], o["src_prev"] = 1, -1, 0
        level0 = (max(o["level"] for o in sched) + 1) if sched else level0
        out.extend(sched)
    model.ops = out
    model.exchange = [(int(lo), int(hi)) for lo, hi in merged]
    model.shard = (int(rank), int(world))
    info = getattr(model, "stage_info", None)
    if info is not None:
        info["enabled"] = False


def _overlap(a, b):
    if a[0] != b[0]:
        return False
    if a[0] == "b":
        return a[1] == b[1]
    return a[1] < b[2] and b[1] < a[2]


def is_micro(o):
    k = o["kind"]
    if k in ("fill", "table", "axpy", "lowpass", "gate", "lincomb"):
        return True
    return k == "matvec" and o["cols"] <= 16 and o["rows"] <= 8192


def schedule_ops(ops, model):
    """Order by *sets -> incs -> reads -> updates* per resource; returns ops annotated with ``level``.

    Kahn's algorithm in rounds.  A round takes every ready cheap vector op ("micro") at once - they
    were all ready together, so they touch disjoint data and need no barrier between them - or,
    when none is ready, every ready big op (oldest first: independent branches of the graph - the four
    dense ensembles of a SLAM network, say - advance side by side, so their small follow-up operators
    become ready together and share one launch).  ``level`` is the round number: the device-side
    program executor puts a workgroup barrier between micro ops of different level, and runs of
    consecutive micro ops become one launch.
    """
    n = len(ops)
    acc = [op_access(o, model) for o in ops]
    entries = [(res, cls, i) for i, a in enumerate(acc) for cls in range(4) for res in a[cls]]
    succ = [set() for _ in range(n)]
    for x in range(len(entries)):
        rx, cx, ix = entries[x]
        for y in range(x + 1, len(entries)):
            ry, cy, iy = entries[y]
            if ix == iy or (cx == 2 and cy == 2) or not _overlap(rx, ry):
                continue
            if cx == cy:      # same class on overlapping data: keep creation order (deterministic sums)
                a, b = (ix, iy) if ops[ix]["seq"] <= ops[iy]["seq"] else (iy, ix)
            else:
                a, b = (ix, iy) if cx < cy else (iy, ix)
            succ[a].add(b)
    indeg = [0] * n
    for a in range(n):
        for b in succ[a]:
            indeg[b] += 1
    ready = [i for i in range(n) if indeg[i] == 0]
    out, level = [], 0
    while ready:
        micro = [i for i in ready if is_micro(ops[i])]
        # big ops of a round: grouped by kind (then oldest first) so that the device can batch neighbours of one kind
        pick = (sorted(micro, key=lambda i: ops[i]["seq"]) if micro else
                sorted(ready, key=lambda i: (ops[i]["kind"], 1 if ops[i].get("dft") else 0, ops[i]["seq"])))
        for i in pick:
            ready.remove(i)
        for i in pick:
            o = dict(ops[i])
            o["level"], o["micro"] = level, is_micro(o)
            out.append(o)
            for j in succ[i]:
                indeg[j] -= 1
                if indeg[j] == 0:
                    ready.append(j)
        level += 1
    if len(out) != n:
        raise fe.BuildError("operator graph has a cycle within one timestep (a loop without a synapse)")
    return out


def build(network, dt=0.001, seed=None, n_eval_points=None, solver_backend="auto", vco_shard=None,
          probes=None, prune=False, staged=True, neuron_shard=None, replicate=(), collapse=True):
    """Build ``network`` into a :class:`BuiltModel`.  ``staged=False`` keeps every operator in the
    per-timestep core (no time-batched pre/post stages); ``collapse=False`` keeps nengo's one-operator-per-connection
    glue (no lincomb folding, glue.py)."""
    return Builder(network, dt=dt, seed=seed, n_eval_points=n_eval_points, solver_backend=solver_backend,
                   vco_shard=vco_shard, probes=probes, prune=prune, staged=staged, neuron_shard=neuron_shard,
                   replicate=replicate, collapse=collapse).build()
