"""``Simulator``: the ``nengo.Simulator`` API on top of the HIP step loop (libssn_hip.so).

Mirrors the simulator surface the reference's scripts use (SURVEY §8b;
``experiments/run_pathint.py:147-165,171-181``, ``experiments/run_slam.py:198-235,250-268``):

    sim = Simulator(model, dt=0.001)        # build + upload + plan + capture the step graph
    with sim:
        sim.run(T)                          # == run_steps(round(T/dt))
    sim.data[probe]                         # (n_samples, size) float64 ndarray
    sim.trange()                            # dt * arange(1, n_steps + 1)
    sim.data[ensemble].gain / .bias / .encoders / .scaled_encoders / ...

Python never runs inside the step loop: Nodes that are functions of ``t`` only are evaluated once
per step *before* the run and uploaded as tables (``prepare``); function nodes with inputs must map
to a kernel (``node.native``).  Errors from the library surface as ``SimulationError`` /
``BuildError`` (nengo's exception names).
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib
from . import frontend as fe
from .builder import BuiltModel, build


class SimulationData(dict):
    """``sim.data``: probes map to sample arrays, model objects to their built parameters."""

    def __init__(self, sim):
        super().__init__()
        self._sim = sim

    def __getitem__(self, key):
        sim = self._sim
        if key in sim._probe_index:
            return sim._probe_array(key)
        if key in sim.model.params:
            return sim.model.params[key]
        raise KeyError(key)

    def __contains__(self, key):
        return key in self._sim._probe_index or key in self._sim.model.params


def pack_model(model, dtype, device=0, steps_per_graph=0, block_steps=0, flags=0):
    """BuiltModel -> (ssn_model_desc, keep-alive list) for ``ssn_create``."""
    keep = []
    sig_init = np.ascontiguousarray(model.sig_init, dtype=np.float64)
    keep.append(sig_init)
    bufs = (_lib.BufferDesc * max(1, len(model.buffers)))()
    for i, b in enumerate(model.buffers):
        if b.dtype.kind in "iu":
            arr = np.ascontiguousarray(b, dtype=np.int32)
            kind = _lib.SSN_BUF_I32
        else:
            arr = np.ascontiguousarray(b, dtype=np.float64)
            kind = _lib.SSN_BUF_REAL
        keep.append(arr)
        bufs[i].data = arr.ctypes.data
        bufs[i].count = arr.size
        bufs[i].kind = kind
    ops = (_lib.OpDesc * len(model.ops))()
    for j, o in enumerate(model.ops):
        k = o["kind"]
        d = ops[j]
        d.kind = _lib.OP_CODE[k]
        d.level = int(o.get("level", 0))
        d.stage = int(o.get("stage", 1))
        d.border = int(o.get("border", -1))
        d.src_prev = int(o.get("src_prev", 0))
        d.phase = int(o.get("phase", 0))
        ii, ff = [0] * 12, [0.0] * 4
        if k == "fill":
            ii[:2] = [o["dst"], o["len"]]
            ff[0] = o["value"]
        elif k == "table":
            ii[:3] = [o["dst"], o["width"], o["table"]]
        elif k == "axpy":
            ii[:4] = [o["dst"], o["src"], o["len"], 1 if o["mode"] == "set" else 0]
            ff[0] = o["alpha"]
        elif k == "lincomb":
            ii[:5] = [o["dst"], o["len"], len(o["srcs"]), o["srcs_buf"], o["alphas_buf"]]
            ff[:2] = [o["self"], o["const"]]
        elif k == "matvec":
            ii[:7] = [o["dst"], o["src"], o["rows"], o["cols"], o["w"], 1 if o["mode"] == "set" else 0, int(o.get("dft", 0) or 0)]
        elif k == "lowpass":
            ii[:3] = [o["dst"], o["src"], o["len"]]
            ff[:2] = [o["a"], o["gain"]]
        elif k == "ensarray":
            nd = o["neuron"]
            ii[:12] = [o["x"], o["K"], o["n"], o["din"], o["dout"], o["enc"], o["bias"], o["dec"], o["dst_idx"],
                       o["v"], o["r"], _lib.NEURON_CODE[nd["type"]]]
            ff[:3] = [nd["tau_rc"], nd["tau_ref"], nd["min_voltage"]]
        elif k == "neurons":
            nd = o["neuron"]
            ii[:6] = [o["j"], o["out"], o["n"], o["v"], o["r"], _lib.NEURON_CODE[nd["type"]]]
            ff[:4] = [nd["tau_rc"], nd["tau_ref"], nd["min_voltage"], o["amp"]]
        elif k == "pes":
            ii[:5] = [o["w"], o["rows"], o["cols"], o["err"], o["act"]]
            ff[0] = o["kappa"]
        elif k == "voja":
            ii[:7] = [o["w"], o["rows"], o["cols"], o["spk"], o["key"], o["learn"], o["scale_buf"]]
            ff[0] = o["lr_dt"]
        elif k == "cleanup":
            ii[:5] = [o["dst"], o["src"], o["rows"], o["cols"], o["w"]]
            if "g_dft" in o:      # factor tables of the sample grid (buffer ids + 1; 0 = table only)
                ii[5:11] = [o["g_dft"] + 1, o["g_lhs"] + 1, o["g_rhs"] + 1, o["grid_rows"], o["grid_cols"], o["grid_k2"]]
        elif k == "gate":
            ii[:3] = [o["dst"], o["src"], o["d"]]
            ff[:2] = [o["thres"], o["rate"]]
        else:
            raise fe.BuildError(f"operator {k!r} has no device encoding")
        for a in range(12):
            d.i[a] = int(ii[a])
        for a in range(4):
            d.f[a] = float(ff[a])
    sig_probes = [p for p in model.probes if "src" in p]
    probes = (_lib.ProbeDesc * max(1, len(sig_probes)))()
    info = getattr(model, "stage_info", None) or {}
    pstage = info.get("probe_stage")
    all_idx = [i for i, p in enumerate(model.probes) if "src" in p]
    for j, p in enumerate(sig_probes):
        probes[j].src, probes[j].width, probes[j].every = int(p["src"]), int(p["width"]), int(p["every"])
        probes[j].stage = int(pstage[all_idx[j]]) if pstage is not None else 1
    p2c, c2p = info.get("pre_to_core", []), info.get("core_to_post", [])
    r_p2c = (_lib.Range * max(1, len(p2c)))()
    r_c2p = (_lib.Range * max(1, len(c2p)))()
    for j, (lo, hi) in enumerate(p2c):
        r_p2c[j].lo, r_p2c[j].hi = int(lo), int(hi)
    for j, (lo, hi) in enumerate(c2p):
        r_c2p[j].lo, r_c2p[j].hi = int(lo), int(hi)
    desc = _lib.ModelDesc()
    desc.abi_version = _lib.SSN_ABI_VERSION
    desc.dtype = _lib.SSN_F64 if dtype in ("f64", "float64", np.float64) else _lib.SSN_F32
    desc.device = int(device)
    desc.n_tables = len(model.tables)
    desc.dt = model.dt
    desc.n_signals = model.sig_size
    desc.signal_init = sig_init.ctypes.data_as(C.POINTER(C.c_double))
    desc.n_buffers, desc.n_ops, desc.n_probes = len(model.buffers), len(model.ops), len(sig_probes)
    desc.steps_per_graph = int(steps_per_graph)
    desc.buffers, desc.ops, desc.probes = bufs, ops, probes
    desc.n_pre_to_core, desc.n_core_to_post = len(p2c), len(c2p)
    desc.pre_to_core, desc.core_to_post = r_p2c, r_c2p
    desc.block_steps = int(block_steps)
    desc.flags = int(flags)
    xr = getattr(model, "exchange", None) or []
    r_x = (_lib.Range * max(1, len(xr)))()
    for j, (lo, hi) in enumerate(xr):
        r_x[j].lo, r_x[j].hi = int(lo), int(hi)
    desc.n_exchange, desc.exchange = len(xr), r_x
    keep += [bufs, ops, probes, r_p2c, r_c2p, r_x]
    return desc, keep, sig_probes


class RowStage:
    """Reusable host staging for the rows of one tabulated node (a fresh 160 MB array per chunk would spend longer in page
    faults than the closures take to evaluate)."""

    def __init__(self):
        self.buf = None

    def rows(self, n, width):
        if self.buf is None or self.buf.shape[0] < n or self.buf.shape[1] != width:
            self.buf = np.empty((max(n, 2 * (0 if self.buf is None or self.buf.shape[1] != width else self.buf.shape[0])), width))
        return self.buf[:n]


def tabulate(fn, width, steps, dt, stage=None):
    """Evaluate a t-only node function for 1-based step numbers ``steps`` with nengo's time
    ``t = step*dt`` (float64) and run-length-encode equal consecutive rows
    (``idx[j]`` = row of step ``steps[j]``; -1 = a row of zeros, which is not stored).

    A plain closure - the reference scripts' ``lambda t: table[int((t - dt) / dt)]``, ``run_pathint.py:134-136`` - is
    called once per timestep in time order, exactly as nengo calls it, and its value is copied before the next call (a
    closure may hand out one buffer again and again); everything else happens in bulk: neighbours are compared bit for bit
    only where three sampled columns agree (runs of candidates as slices, no gathers), and the zero test runs on the
    surviving rows.  The returned rows may alias ``stage`` (valid until the next call with the same stage)."""
    if hasattr(fn, "table"):          # vectorised provider: (rows, idx) for all steps at once
        rows, idx = fn.table(np.asarray(steps))
        return np.asarray(rows, dtype=np.float64).reshape(-1, width), np.asarray(idx, dtype=np.int32)
    ts = (np.asarray(steps) * dt).tolist()                            # t = step*dt in float64, as nengo computes it
    n = len(ts)
    if n == 0:
        return np.zeros((0, width)), np.zeros(0, dtype=np.int32)
    rows = (stage or RowStage()).rows(n, width)
    for j, t in ((0, ts[0]), (n - 1, ts[-1])):                        # (sizes are checked on the first and the last value:
        if np.size(fn(t)) != width:                                   #  an assignment would broadcast a scalar silently)
            raise fe.SimulationError(f"node function returned {np.size(fn(t))} values, expected {width}")
    try:
        for j, t in enumerate(ts):                                    # ONE Python call per timestep, in time order
            rows[j] = fn(t)
    except ValueError as e:
        raise fe.SimulationError(f"node function returned a value that is not {width} wide") from e
    bits = rows.view(np.uint64)                                       # bitwise row comparison (-0.0 is not 0.0, NaN equals itself)
    cols = sorted({0, width // 2, width - 1})

    def runs(mask):                                                   # [lo, hi) of every run of True
        edge = np.flatnonzero(np.diff(np.concatenate(([False], mask, [False])).astype(np.int8)))
        return zip(edge[0::2].tolist(), edge[1::2].tolist())

    change = np.ones(n, dtype=bool)
    if n > 1:
        same = np.ones(n - 1, dtype=bool)
        for c in cols:
            same &= bits[1:, c] == bits[:-1, c]
        for lo, hi in runs(same):          # neighbours that agree on the sampled columns: compared in full, slice against slice
            same[lo:hi] = (bits[lo + 1:hi + 1] == bits[lo:hi]).all(axis=1)
        change[1:] = ~same
    zero = np.ones(n, dtype=bool)
    for c in cols:
        zero &= bits[:, c] == 0
    for lo, hi in runs(zero):              # rows whose sampled columns are zero: tested in full
        zero[lo:hi] = ~bits[lo:hi].any(axis=1)
    first = np.flatnonzero(change)                                    # first timestep of every run of equal rows
    run = np.cumsum(change) - 1                                       # run of every timestep
    if 2 * first.size >= n:
        # mostly distinct rows (a path table; int((t - dt) / dt) repeats a row now and then): the staged rows are the table as
        # they stand - a run points at its first row, the repeats in between are dead weight of the upload, nothing is copied
        slot = first.copy()
        slot[zero[first]] = -1
        return rows, slot[run].astype(np.int32)
    z = zero[first]
    slot = np.cumsum(~z) - 1                                          # stored position of every non-zero run's row
    slot[z] = -1
    return np.ascontiguousarray(rows[first[~z]]), slot[run].astype(np.int32)


class Simulator:
    def __init__(self, network, dt=0.001, seed=None, progress_bar=None, dtype="f32", device=0,
                 n_eval_points=None, steps_per_graph=0, model=None, vco_shard=None, block_steps=0, flags=0):
        self.dt = float(dt)
        self.closed = True
        self._lib = _lib.load()          # fails loudly when the HIP library is not built
        if model is None:
            if isinstance(network, BuiltModel):
                model = network
            else:
                model = build(network, dt=dt, seed=seed, n_eval_points=n_eval_points, vco_shard=vco_shard)
        self.model = model
        self.dtype = "f64" if dtype in ("f64", "float64", np.float64) else "f32"
        self._block_align = int(block_steps) if block_steps else 1024      # (the library's default block of a staged model)
        desc, keep, self._sig_probes = pack_model(model, self.dtype, device, steps_per_graph, block_steps, flags)
        self._h = C.c_void_p()
        t0 = time.time()
        self._check(self._lib.ssn_create(C.byref(desc), C.byref(self._h)), build=True)
        self.upload_seconds = time.time() - t0
        del keep
        self.closed = False
        self._probe_index = {}
        self._chunks = {}
        for j, p in enumerate(self._sig_probes):
            self._probe_index[p["probe"]] = ("sig", j, p)
        for p in model.probes:
            if "src" not in p:
                self._probe_index[p["probe"]] = ("buf", None, p)
        for key in self._probe_index:
            self._chunks[key] = []
        self.data = SimulationData(self)
        self._fetched = {}
        self._prepared_until = 0
        self.n_steps = 0
        # SSN_TRACE_RUN=1: (phase, start, end, detail) of every host-side phase of a pipelined run, perf_counter seconds
        # (tools/bench_end_to_end.py prints the timeline)
        self.trace = [] if os.environ.get("SSN_TRACE_RUN") else None

    def _tr(self, name, t0, detail=None):
        if self.trace is not None:
            self.trace.append((name, t0, time.perf_counter(), detail))

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc, build=False):
        if rc != 0:
            msg = f"{_lib.STATUS.get(rc, rc)}: {_lib.last_error()}"
            raise (fe.BuildError if build else fe.SimulationError)(msg)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if not self.closed:
            self._lib.ssn_destroy(self._h)
            self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def time(self):
        return self.n_steps * self.dt

    def trange(self, sample_every=None):
        every = 1 if sample_every is None else max(1, int(round(sample_every / self.dt)))
        return self.dt * every * np.arange(1, self.n_steps // every + 1)

    # -- inputs --------------------------------------------------------------------------------
    def prepare(self, n_steps):
        """Tabulate every t-only node for the next ``n_steps`` steps, upload the tables and reserve
        probe storage, so that ``run_steps(n_steps)`` starts with all inputs resident in HBM."""
        n_steps = int(n_steps)
        if getattr(self, "_uncollected", False):
            self._collect()              # a new reservation drops device-side samples: fetch them first
        first = self.n_steps
        steps = np.arange(first + 1, first + n_steps + 1)
        for tid, tb in enumerate(self.model.tables):
            rows, idx = tabulate(tb["fn"], tb["width"], steps, self.dt)
            rows = np.ascontiguousarray(rows, dtype=np.float64)
            self._check(self._lib.ssn_set_table(self._h, tid, rows.ctypes.data, rows.shape[0], tb["width"],
                                                idx.ctypes.data, idx.size, first))
        self._check(self._lib.ssn_reserve_probes(self._h, n_steps))
        self._fetched = {}               # sample counts restart with the new reservation
        self._prepared_until = first + n_steps

    def reserve_probes(self, n_steps):
        """Probe storage for the next ``n_steps`` steps (drops device-side samples: they are fetched first)."""
        if getattr(self, "_uncollected", False):
            self._collect()
        self._check(self._lib.ssn_reserve_probes(self._h, int(n_steps)))
        self._fetched = {}
        self._reserved_until = self.n_steps + int(n_steps)

    def prepare_tables_device(self, tables, n_steps, reserve=True):
        """Like ``prepare`` for inputs that are already in HBM (multi-GPU exchange): ``tables`` maps table id ->
        (device pointer to rows [n_rows][width] in the simulator's dtype, n_rows, int32 row index per step).
        ``reserve=False``: probe storage was reserved for a longer stretch by ``reserve_probes``."""
        n_steps = int(n_steps)
        first = self.n_steps
        if not reserve and getattr(self, "_reserved_until", 0) >= first + n_steps:
            for tid, (ptr, n_rows, idx) in tables.items():
                self.set_table_device(tid, ptr, n_rows, idx, first)
            self._prepared_until = first + n_steps
            return
        if getattr(self, "_uncollected", False):
            self._collect()
        if set(tables) != set(range(len(self.model.tables))):
            raise fe.SimulationError("prepare_tables_device needs every table of the model")
        for tid, (ptr, n_rows, idx) in tables.items():
            self.set_table_device(tid, ptr, n_rows, idx, first)
        self._check(self._lib.ssn_reserve_probes(self._h, n_steps))
        self._fetched = {}
        self._prepared_until = first + n_steps

    def probe_count(self, probe):
        """Samples of a signal probe held on the device in the current reservation."""
        kind, j, p = self._probe_index[probe]
        return int(self._lib.ssn_probe_count(self._h, j))

    def set_table_device(self, table_id, rows_dev_ptr, n_rows, idx, first_step):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        tb = self.model.tables[table_id]
        self._check(self._lib.ssn_set_table_device(self._h, table_id, C.c_void_p(rows_dev_ptr), n_rows, tb["width"],
                                                   idx.ctypes.data, idx.size, first_step))

    # A long run straight from run() is cut into chunks: the node closures of chunk k + 1 are evaluated on a helper thread and
    # the samples of chunk k - 1 are read back on another while the device steps chunk k.  What is NOT hidden: the tabulation
    # of the first chunk, the read-back of the last one, and ~0.3 ms of idle device at every boundary.  So the first chunk is
    # short, every later one is as long as its tabulation can hide behind the chunk before it (measured rates, at most
    # PIPELINE_GROWTH x its predecessor), and the run ends with a short chunk.
    PIPELINE_FIRST = 512
    PIPELINE_MIN = 256
    PIPELINE_MAX = 16384
    PIPELINE_TAIL = 512
    PIPELINE_MID = 2048
    PIPELINE_GROWTH = 4.0
    PIPELINE_CHUNK = 2048            # runs of at most twice this many steps are prepared in one piece

    def _tabulate_chunk(self, first, n):
        """Tabulate every t-only node for the n timesteps after 0-based step ``first`` and STAGE the tables on the device
        (``ssn_stage_table``: DMA into a second set of buffers, legal while another thread's ``ssn_run_steps`` is in flight -
        this runs on the helper thread of a pipelined run); ``ssn_commit_tables`` between two chunks makes them current.
        Returns (first, n)."""
        steps = np.arange(first + 1, first + n + 1)
        if self._stages is None:
            self._stages = [RowStage() for _ in self.model.tables]
        t_np = np.float64 if self.dtype == "f64" else np.float32
        for tid, (tb, stage) in enumerate(zip(self.model.tables, self._stages)):
            # (the stage is free again: the call below copies the rows before it returns)
            rows, idx = tabulate(tb["fn"], tb["width"], steps, self.dt, stage)
            rows = np.ascontiguousarray(rows, dtype=t_np)              # the simulator's own type: the upload is a plain copy
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            self._check(self._lib.ssn_stage_table(self._h, tid, rows.ctypes.data, rows.shape[0], tb["width"],
                                                  idx.ctypes.data, idx.size, first))
        return first, n

    _stages = None

    # -- running -------------------------------------------------------------------------------
    def run(self, time_in_seconds, progress_bar=None):
        self.run_steps(int(np.round(float(time_in_seconds) / self.dt)))

    def step(self):
        self.run_steps(1)

    def run_steps(self, steps, profile=False, collect=True):
        steps = int(steps)
        if steps <= 0:
            return
        if self.closed:
            raise fe.SimulationError("simulator is closed")
        pipelined = None
        if self._prepared_until < self.n_steps + steps or self._prepared_until == 0:
            if steps > 2 * self.PIPELINE_CHUNK and self.model.tables and not profile:
                # long run straight from run(): the node closures of chunk k+1 are evaluated on a helper thread while
                # the device steps chunk k (ssn_run_steps releases the GIL); tables are uploaded between chunks
                if getattr(self, "_uncollected", False):
                    self._collect()
                t_tab = time.perf_counter()
                self._check(self._lib.ssn_reserve_probes(self._h, steps))
                self._tr("reserve", t_tab)
                self._fetched = {}
                self._reserved_until = self.n_steps + steps
                t_tab = time.perf_counter()
                pipelined = self._tabulate_chunk(self.n_steps, min(self.PIPELINE_FIRST, steps))
                self._tr("tabulate first", t_tab, pipelined[1])
                self._tab_rate = (time.perf_counter() - t_tab) / max(1, pipelined[1])      # seconds per timestep, host
                self._dev_rate = None                                                        # ... device: known after the first chunk
                if collect:
                    # the samples of chunk k are fetched (float64, straight into one array per probe) on a helper thread
                    # while the device steps chunk k + 1: the library downloads on its own stream
                    self._bulk = {}
                    for key, (kind, j, p) in self._probe_index.items():
                        if kind == "sig":
                            cap = (self.n_steps + steps) // p["every"] - self.n_steps // p["every"]
                            self._bulk[key] = np.empty((cap, p["width"]), dtype=np.float64)
            else:
                self.prepare(steps)
        buf_probes = [(k, v[2]) for k, v in self._probe_index.items() if v[0] == "buf"]
        self._collector = self._tab_worker = None
        try:
            self._step_loop(steps, profile, pipelined, buf_probes)
        except BaseException:
            # a failed run must not leave the helper threads or the bulk arrays of the pipelined read-back behind: the
            # tabulation thread would go on calling the user's node closures (and this object) after the caller has seen the
            # exception, and the next run_steps would collect into stale arrays
            if self._tab_go is not None:
                self._tab_go.set()               # (a helper still waiting for its start signal)
            for th in (self._tab_worker, self._collector):
                if th is not None:
                    th.join()
            self._collector = self._tab_worker = None
            self._bulk, self._bulk_error = None, None
            raise
        self._uncollected = True
        if collect:
            self._collect()
        self._prepared_until = max(self._prepared_until, self.n_steps)

    _collector = None
    _tab_worker = None
    _tab_go = None
    _tab_gate_off = os.environ.get("SSN_TAB_GATE", "1") == "0"      # A/B knob: the helper starts in front of the device call again
    _tab_rate = None
    _dev_rate = None

    def _next_chunk_len(self, cur, remaining):
        """Length of the chunk after one of ``cur`` timesteps (``remaining`` timesteps are left behind that one)."""
        grow = self.PIPELINE_GROWTH
        if self._dev_rate and self._tab_rate:            # its tabulation runs while the device steps `cur` timesteps
            grow = min(grow, 0.8 * self._dev_rate / max(self._tab_rate, 1e-9))
        n = int(max(self.PIPELINE_MIN, min(self.PIPELINE_MAX, cur * grow)))
        align = getattr(self, "_block_align", 1024)
        if n >= align:
            n -= n % align                               # whole time-batched blocks where the chunk holds several
        # The run's last samples are read back with nothing to hide behind, and the read-back of a long chunk (2.9 ms for 8 736
        # samples of 1015) outlasts a short last chunk: the run ends ... long, PIPELINE_MID, PIPELINE_TAIL - each read-back fits
        # under the chunk after it.
        tail, mid = self.PIPELINE_TAIL, (2 * align if 512 <= align <= 2048 else self.PIPELINE_MID)      # (whole blocks again)
        if remaining <= tail:
            n = remaining
        elif remaining <= mid + tail:
            n = remaining - tail
        elif remaining <= n + mid + tail:
            n = remaining - mid - tail if remaining - mid - tail >= self.PIPELINE_MIN else remaining - tail
        return max(1, min(n, remaining))

    def _step_loop(self, steps, profile, pipelined, buf_probes):
        done = 0
        collector = None
        while done < steps:
            # Probes of learned signals ("weights", "scaled_encoders"): nengo adds a learning rule's delta to its target at
            # the START of the next timestep (`target += delta` is an inc, the delta an update - SURVEY Appendix A.7 / A.8),
            # so the sample of timestep t holds the deltas of timesteps < t.  k_pes / k_voja add theirs at once: the buffer
            # is therefore read BEFORE the timestep whose sample is due.
            for key, p in buf_probes:
                if (self.n_steps + 1) % p["every"] == 0:
                    self._chunks[key].append(self.read_buffer(self._probe_buffer_id(p))[None])
            chunk = steps - done
            worker = None
            if pipelined is not None:
                first, n_tab = pipelined
                t_up = time.perf_counter()
                self._check(self._lib.ssn_commit_tables(self._h))      # (staged by _tabulate_chunk while the chunk before ran)
                self._tr("commit tables", t_up, n_tab)
                self._prepared_until = first + n_tab
                chunk = min(chunk, n_tab)
                nxt = self.n_steps + chunk
                pipelined = None
                if done + chunk < steps:
                    import threading
                    box = {}
                    n_next = self._next_chunk_len(chunk, steps - done - chunk)
                    go = self._tab_go = threading.Event()
                    def tab(box=box, nxt=nxt, n_next=n_next, go=go):
                        go.wait()            # (see below: the device run of this chunk is enqueued first)
                        try:
                            t0 = time.perf_counter()
                            box.update(r=self._tabulate_chunk(nxt, n_next))
                            box["t"] = (time.perf_counter() - t0) / n_next
                        except BaseException as e:       # noqa: BLE001 - re-raised on the caller's thread
                            box["e"] = e
                    worker = self._tab_worker = threading.Thread(target=tab)
                    if self._tab_gate_off:
                        go.set()
                    worker.start()
            for _, p in buf_probes:          # stop before the next timestep whose sample is due (taken at the top of the loop)
                r = (self.n_steps + 1) % p["every"]
                chunk = min(chunk, p["every"] - r if r else p["every"])
            t_dev = time.perf_counter()
            if worker is not None and not self._tab_gate_off:
                # A helper that starts calling plain Python closures keeps the interpreter lock for a whole switch interval (5 ms):
                # started in front of the call below it delayed the launch of this chunk by 4 ms (timeline of round 4).  It waits on
                # an event instead, set right in front of the foreign call - which gives the lock up by itself.
                go.set()
            self._check(self._lib.ssn_run_steps(self._h, chunk, int(profile)))
            self._tr("run", t_dev, chunk)
            if chunk >= self.PIPELINE_MIN:
                self._dev_rate = (time.perf_counter() - t_dev) / chunk
            self.n_steps += chunk
            done += chunk
            if getattr(self, "_bulk", None) is not None and done < steps:
                import threading
                if collector is not None:
                    t_j = time.perf_counter()
                    collector.join()
                    self._tr("wait for read-back", t_j)
                collector = self._collector = threading.Thread(target=self._collect_bulk)
                collector.start()
            if worker is not None:
                t_j = time.perf_counter()
                worker.join()
                self._tr("wait for tabulation", t_j)
                self._tab_worker = None
                if "r" not in box:
                    raise fe.SimulationError("evaluating the input nodes for the next chunk failed") from box.get("e")
                pipelined = box["r"]
                self._tab_rate = box.get("t", self._tab_rate)
                if self.n_steps != pipelined[0]:          # a weight-probe boundary cut the chunk short: re-tabulate from here
                    pipelined = self._tabulate_chunk(self.n_steps, min(self.PIPELINE_FIRST, steps - done))
        if collector is not None:
            t_j = time.perf_counter()
            collector.join()
            self._tr("wait for read-back", t_j)
        self._collector = None
        if getattr(self, "_bulk", None) is not None:
            self._collect_bulk()                     # the last chunk
            if self._bulk_error is not None:
                err, self._bulk_error = self._bulk_error, None
                self._bulk = None
                raise err
            for key, arr in self._bulk.items():
                n = self._fetched.get(key, 0)
                if n:
                    self._chunks[key].append(arr[:n])
            self._bulk = None

    def _probe_buffer_id(self, p):
        b = p["buf"]
        if isinstance(b, tuple):
            return self.model.params[p["ens"]].encoder_buffer
        return b

    _bulk = None
    _bulk_error = None

    def _collect_bulk(self):
        """Fetch the samples completed so far into the per-probe arrays of a pipelined run (helper thread)."""
        t_c = time.perf_counter()
        try:
            for key, arr in self._bulk.items():
                kind, j, p = self._probe_index[key]
                n = min(int(self._lib.ssn_probe_count(self._h, j)), arr.shape[0])
                have = self._fetched.get(key, 0)
                if n <= have:
                    continue
                self._check(self._lib.ssn_read_probe(self._h, j, arr[have:].ctypes.data, have, n - have))
                self._fetched[key] = n
        except BaseException as e:               # noqa: BLE001 - surfaced by run_steps on the caller's thread
            self._bulk_error = e
        self._tr("read-back", t_c)

    def _collect(self):
        """Fetch the probe samples produced since the last fetch (incremental within a reservation)."""
        self._uncollected = False
        for key, (kind, j, p) in self._probe_index.items():
            if kind != "sig":
                continue
            n = int(self._lib.ssn_probe_count(self._h, j))
            have = self._fetched.get(key, 0)
            if n <= have:
                continue
            out = np.empty((n - have, p["width"]), dtype=np.float64)
            self._check(self._lib.ssn_read_probe(self._h, j, out.ctypes.data, have, n - have))
            self._chunks[key].append(out)
            self._fetched[key] = n

    def _probe_array(self, key):
        if getattr(self, "_uncollected", False) and not self.closed:
            self._collect()              # run_steps(collect=False) left samples on the device
        kind, j, p = self._probe_index[key]
        chunks = [c[1] if isinstance(c, tuple) else c for c in self._chunks[key]]
        if not chunks:
            width = p.get("width")
            return np.zeros((0, width)) if width else np.zeros((0,) + tuple(p["shape"]))
        if len(chunks) == 1:
            out = chunks[0]                  # (a pipelined run fetched straight into one array: no 80 MB copy)
        else:
            out = np.concatenate(chunks, axis=0)
            self._chunks[key] = [out]        # later reads of sim.data[probe] reuse it
        view = out.view()
        view.setflags(write=False)           # sim.data hands out read-only views of the simulator's storage, as nengo's does
        return view

    def probe_tail(self, key, n):
        """The last ``n`` samples of a probe without concatenating the whole history."""
        out, need = [], int(n)
        for c in reversed(self._chunks[key]):
            a = c[1] if isinstance(c, tuple) else c
            out.append(a[-need:] if need < a.shape[0] else a)
            need -= out[-1].shape[0]
            if need <= 0:
                break
        return np.concatenate(out[::-1], axis=0) if out else np.zeros((0, self._probe_index[key][2]["width"]))

    def clear_probe_data(self):
        """Forget samples already handed out (long sharded runs stream blocks through)."""
        for k in self._chunks:
            self._chunks[k] = []

    # -- state access ----------------------------------------------------------------------------
    def read_signal(self, off, count):
        out = np.empty(count, dtype=np.float64)
        self._check(self._lib.ssn_read_signal(self._h, off, count, out.ctypes.data))
        return out

    def write_signal(self, off, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self._lib.ssn_write_signal(self._h, off, v.size, v.ctypes.data))

    def read_buffer(self, buffer_id):
        shape = self.model.buffers[buffer_id].shape
        out = np.empty(shape, dtype=np.float64)
        self._check(self._lib.ssn_read_buffer(self._h, buffer_id, out.ctypes.data, out.size))
        return out

    def write_buffer(self, buffer_id, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self._lib.ssn_write_buffer(self._h, buffer_id, v.ctypes.data, v.size))

    def read_probe_device(self, probe, dst_ptr, first, count):
        kind, j, p = self._probe_index[probe]
        self._check(self._lib.ssn_read_probe_device(self._h, j, C.c_void_p(dst_ptr), first, count))

    def counters(self):
        c = _lib.Counters()
        self._check(self._lib.ssn_get_counters(self._h, C.byref(c)))
        return {f: getattr(c, f) for f, _ in c._fields_}

    # -- neuron-sharded models (builder.shard_phases): the caller completes the partial sums between the phases ---------
    def run_phase(self, phase):
        """0: up to the exchange; 1: the updates; 2: the updates followed by the next timestep up to its exchange.
        (The graphs of the stream-ordered path are built through ``phase_async(-1, buffer, None)``.)"""
        if phase == 3:
            self._cycle_segment(lambda: self._lib.ssn_run_phase(self._h, 3))
            return
        if phase not in (0, 1, 2):
            raise fe.SimulationError(f"run_phase({phase}): 0, 1, 2 or 3")
        if phase in (0, 2) and self._prepared_until < self.n_steps + (1 if phase == 0 else 2):
            raise fe.SimulationError("neuron-sharded model: call prepare(n_steps) before stepping")
        self._check(self._lib.ssn_run_phase(self._h, int(phase)))
        if phase >= 1:
            self.n_steps += 1
            self._uncollected = True

    def cycle_steps(self):
        """Timesteps of the plan pipelined over the exchange (0: the model has none): ``cycle_steps() + 1`` calls of
        ``run_phase(3)`` / ``phase_async(3, ...)`` with the caller's exchange between consecutive ones advance that many."""
        return int(self._lib.ssn_cycle_steps(self._h))

    def _cycle_segment(self, call):
        c = self.cycle_steps()
        seg = getattr(self, "_cycle_seg", 0)
        if seg == 0 and self._prepared_until < self.n_steps + c:
            raise fe.SimulationError("neuron-sharded model: call prepare(n_steps) before stepping")
        self._check(call())
        seg += 1
        if seg > c:
            seg = 0
            self.n_steps += c
            self._uncollected = True
        self._cycle_seg = seg

    def phase_async(self, phase, exchange_buf_ptr, stream_ptr):
        """Stream-ordered variant of ``run_phase``: enqueues [unpack] -> the phase -> [pack] as one graph launch on the
        caller's HIP stream and returns; ``phase_sync`` waits and checks.  No host synchronisation per timestep."""
        if phase == -1:                  # only build the graphs for this exchange buffer (no launch, nothing to prepare)
            self._check(self._lib.ssn_phase_async(self._h, -1, C.c_void_p(exchange_buf_ptr or None), C.c_void_p(None)))
            return
        if phase == 3:
            self._cycle_segment(lambda: self._lib.ssn_phase_async(self._h, 3, C.c_void_p(exchange_buf_ptr or None), C.c_void_p(stream_ptr or None)))
            return
        if phase in (0, 2) and self._prepared_until < self.n_steps + (1 if phase == 0 else 2):
            raise fe.SimulationError("neuron-sharded model: call prepare(n_steps) before stepping")
        self._check(self._lib.ssn_phase_async(self._h, int(phase), C.c_void_p(exchange_buf_ptr or None), C.c_void_p(stream_ptr or None)))
        if phase >= 1:
            self.n_steps += 1
            self._uncollected = True

    def phase_sync(self, stream_ptr):
        self._cycle_seg = 0              # (the library leaves a cycle it is synchronised inside of, with an error)
        self._check(self._lib.ssn_phase_sync(self._h, C.c_void_p(stream_ptr or None)))

    def exchange_size(self):
        return int(self._lib.ssn_exchange_size(self._h))

    def exchange_pack(self, dev_ptr):
        self._check(self._lib.ssn_exchange_pack(self._h, C.c_void_p(dev_ptr)))

    def exchange_unpack(self, dev_ptr):
        self._check(self._lib.ssn_exchange_unpack(self._h, C.c_void_p(dev_ptr)))

    def exchange_host(self, allreduce):
        """Host path (gloo): the exchange ranges as one float64 vector -> ``allreduce(vector)`` (in place) -> back."""
        ranges = self.model.exchange
        vec = np.concatenate([self.read_signal(lo, hi - lo) for lo, hi in ranges]) if ranges else np.zeros(0)
        allreduce(vec)
        off = 0
        for lo, hi in ranges:
            self.write_signal(lo, vec[off:off + hi - lo])
            off += hi - lo

    def kernel_times(self):
        """{kernel name: (launches, total ms)} of the launches timed by ``run_steps(..., profile=2)``."""
        arr = (_lib.KernelTime * 32)()
        n = self._lib.ssn_get_kernel_times(self._h, arr, 32)
        if n < 0:
            self._check(n)
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].ms_total)) for i in range(min(n, 32))}

    def reset(self, seed=None):
        self._check(self._lib.ssn_reset(self._h))
        self.n_steps = 0
        self._cycle_seg = 0
        self._prepared_until = 0
        for k in self._chunks:
            self._chunks[k] = []
        self._fetched = {}
