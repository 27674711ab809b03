"""Built-model cache: ``cached_build(network, ...)`` = ``builder.build`` with the frozen BuiltModel kept on disk.

The reference keeps what a run produced (``experiments/run_slam.py:282-293``, ``run_pathint.py:201-207``) but rebuilds its
network - eval points, decoder solves - in every process; at BASELINE configs[2] that is ~60 s of Cholesky solves before the
first timestep (checkpoint row).  A BuiltModel is a pure function of (the network's declaration, the
seeds, the builder's arguments, the builder's code), so it is cached under a fingerprint of exactly those:

* every ensemble / node / connection / probe of the network in declaration order: sizes, labels, seeds, neuron types,
  distributions, given arrays BY CONTENT, slices, synapses, solvers, learning rules, and the ensemble-array grouping;
* functions that end up inside the model - decoder targets of connections (``feedback``, ``np.square``, reference
  ``pathintegration.py:118-125``, ``binding.py:316-317``) and function nodes with inputs (clean-up, gate; recognised by
  behaviour) - by byte code, constants, defaults, closure cells and the globals they name, recursively; functions of ``t``
  alone are NOT part of a built model (they are tabulated at run time from the live network) and are not hashed;
* ``dt``, the resolved root seed, eval points, shard arguments, staging / collapse switches, the decoder solver that
  ``solver_backend`` resolves to on this machine (NumPy vs torch on a GPU round differently);
* the source text of this package (any change to the builder invalidates every entry).

Anything the fingerprint cannot describe deterministically (an unseeded network, an object it cannot walk) makes the build
uncacheable: ``cached_build`` then simply builds.  A hit re-binds the stored model to the objects of the *current* network
(``model.params`` keys, probes, table functions), so the result is interchangeable with a fresh build - bit-identical buffers
(``tests/test_modelcache.py``).

Cache directory: ``$SSN_CACHE_DIR``, else ``$XDG_CACHE_HOME/sspslam_amd``, else ``~/.cache/sspslam_amd``.  ``SSN_NO_CACHE=1``
switches it off.  Entries are written to a temporary name and renamed (two processes may race to fill the same entry).
"""
import functools
import hashlib
import os
import pickle
import time
import types

import numpy as np

from . import frontend as fe
from .builder import BuiltModel, build

FORMAT = 2
_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


class Uncacheable(Exception):
    pass


def cache_dir():
    d = os.environ.get("SSN_CACHE_DIR")
    if not d:
        d = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "sspslam_amd")
    return d


@functools.lru_cache(maxsize=1)
def _source_digest():
    h = hashlib.blake2b(digest_size=16)
    for root, _, files in sorted(os.walk(_PKG_DIR)):
        if "__pycache__" in root or os.sep + "csrc" in root:
            continue
        for f in sorted(files):
            if f.endswith(".py"):
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def _network_objects(network, extra_probes=None):
    """Declaration-order enumeration of everything a BuiltModel may refer to -> {id(obj): (kind, index)}, lists.
    ``extra_probes``: the ``probes=`` override of a build (probes that are not - or no longer - the network's own)."""
    own = list(network.all_probes)
    groups = (("e", list(network.all_ensembles)), ("n", list(network.all_nodes)), ("c", list(network.all_connections)),
              ("p", own), ("x", [p for p in (extra_probes or ()) if not any(p is q for q in own)]),
              ("w", [network] + list(network.all_networks)))
    index = {}
    for kind, objs in groups:
        for i, o in enumerate(objs):
            index[id(o)] = (kind, i)
    return index, dict(groups)


class _Fingerprint:
    """Structural hash of a network declaration (see the module docstring for what goes in)."""

    MAX_DEPTH = 12

    def __init__(self, index):
        self.h = hashlib.blake2b(digest_size=20)
        self.index = index
        self.seen = {}

    def feed(self, *parts):
        for p in parts:
            b = p if isinstance(p, bytes) else str(p).encode()
            self.h.update(len(b).to_bytes(8, "little") + b)

    def ref(self, obj):
        """A network object as a reference (kind, index) - never walked from inside another object."""
        if isinstance(obj, fe.ObjView) or (hasattr(obj, "obj") and hasattr(obj, "slice")):
            idx = getattr(obj, "indices", None)
            self.feed("view")
            self.ref(obj.obj)
            self.value(np.asarray(idx) if idx is not None else repr(obj.slice))
            return
        if isinstance(obj, fe.Neurons):
            self.feed("neurons-of")
            self.ref(obj.ensemble)
            return
        if isinstance(obj, fe.LearningRule):
            self.feed("rule-of")
            self.ref(obj.connection)
            return
        key = self.index.get(id(obj))
        if key is None:
            raise Uncacheable(f"{obj!r} is not part of the network")
        self.feed("ref", key[0], key[1])

    def value(self, v, depth=0):
        if depth > self.MAX_DEPTH:
            raise Uncacheable("object graph too deep to fingerprint")
        if v is None or isinstance(v, (bool, int, float, complex, str, bytes)):
            self.feed(type(v).__name__, repr(v))
        elif isinstance(v, np.ndarray):
            a = np.ascontiguousarray(v)
            if a.dtype == object:
                self.feed("objarray", a.shape)
                for x in a.reshape(-1):
                    self.value(x, depth + 1)
            else:
                self.feed("ndarray", a.shape, a.dtype.str)
                self.h.update(memoryview(a).cast("B") if a.size else b"")
        elif isinstance(v, np.generic):
            self.feed("npscalar", v.dtype.str, repr(v.item()))
        elif isinstance(v, (list, tuple)):
            self.feed(type(v).__name__, len(v))
            for x in v:
                self.value(x, depth + 1)
        elif isinstance(v, (set, frozenset)):
            self.feed("set", len(v))
            for x in sorted(v, key=repr):
                self.value(x, depth + 1)
        elif isinstance(v, dict):
            self.feed("dict", len(v))
            for k in sorted(v, key=repr):
                self.value(k, depth + 1)
                self.value(v[k], depth + 1)
        elif isinstance(v, (slice, range)):
            self.feed(repr(v))
        elif isinstance(v, np.ufunc):
            self.feed("ufunc", v.__name__)
        elif isinstance(v, (np.random.RandomState, np.random.Generator)):
            # a live generator hanging off a captured object (HexagonalSSPSpace keeps the reference's unseeded `rng`,
            # sspspace.py:680): its state is not part of the declaration.  Whatever a build draws from it is not reproducible
            # from ANY description of the network, so the cached model is one of that declaration's legitimate builds.
            self.feed("rng", type(v).__name__)
        elif isinstance(v, types.ModuleType):
            self.feed("module", v.__name__)
        elif id(v) in self.index or isinstance(v, (fe.ObjView, fe.Neurons, fe.LearningRule)):
            self.ref(v)
        elif isinstance(v, type):
            self.feed("class", v.__module__, v.__qualname__)
        elif isinstance(v, (types.FunctionType, types.MethodType, types.BuiltinFunctionType, functools.partial)) or \
                (callable(v) and not hasattr(v, "__dict__")):
            self.function(v, depth + 1)
        elif isinstance(v, types.CodeType):
            self.code(v, depth + 1)
        else:
            # plain objects (distributions, neuron types, synapses, solvers, SSP spaces held by a closure ...): class + attributes
            if id(v) in self.seen:
                self.feed("again", self.seen[id(v)])
                return
            self.seen[id(v)] = len(self.seen)
            self.feed("object", type(v).__module__, type(v).__qualname__)
            d = getattr(v, "__dict__", None)
            if d is None:
                slots = [s for c in type(v).__mro__ for s in getattr(c, "__slots__", ())]
                if not slots:
                    raise Uncacheable(f"cannot describe {type(v).__qualname__}")
                d = {s: getattr(v, s) for s in slots if hasattr(v, s)}
            self.value({k: x for k, x in d.items() if not (isinstance(k, str) and k.startswith("__"))}, depth + 1)
            if callable(v) and hasattr(type(v), "__call__") and isinstance(getattr(type(v), "__call__"), types.FunctionType):
                self.function(type(v).__call__, depth + 1)

    def code(self, co, depth):
        self.feed("code", co.co_name, co.co_argcount, co.co_kwonlyargcount, co.co_flags & 0x0F)
        self.h.update(co.co_code)
        self.value(co.co_names, depth)
        self.value(co.co_varnames, depth)
        for c in co.co_consts:
            self.value(c, depth + 1)

    def function(self, fn, depth):
        if id(fn) in self.seen:
            self.feed("again", self.seen[id(fn)])
            return
        self.seen[id(fn)] = len(self.seen)
        if isinstance(fn, functools.partial):
            self.feed("partial")
            self.value(fn.func, depth + 1)
            self.value(fn.args, depth + 1)
            self.value(fn.keywords, depth + 1)
            return
        if isinstance(fn, types.MethodType):
            self.feed("method")
            self.function(fn.__func__, depth + 1)
            self.value(fn.__self__, depth + 1)
            return
        code = getattr(fn, "__code__", None)
        if code is None:                      # builtins, C functions
            self.feed("builtin", getattr(fn, "__module__", None), getattr(fn, "__qualname__", getattr(fn, "__name__", repr(fn))))
            return
        self.code(code, depth + 1)
        self.value(fn.__defaults__, depth + 1)
        self.value(fn.__kwdefaults__, depth + 1)
        for cell in (fn.__closure__ or ()):
            try:
                self.value(cell.cell_contents, depth + 1)
            except ValueError:                # empty cell
                self.feed("empty-cell")
        g = fn.__globals__

        def names(co):
            out = list(co.co_names)
            for c in co.co_consts:
                if isinstance(c, types.CodeType):
                    out += names(c)
            return out

        for nm in sorted(set(names(code))):
            if nm in g:
                x = g[nm]
                self.feed("global", nm)
                if isinstance(x, types.ModuleType):
                    self.feed("module", x.__name__)
                else:
                    self.value(x, depth + 1)
        # (attributes of a function object - a vectorised `.batch` / `.table` twin the builder may call instead)
        for nm, x in sorted(getattr(fn, "__dict__", {}).items()):
            self.feed("fattr", nm)
            self.value(x, depth + 1)


def _resolved_solver(backend):
    from . import solvers
    torch, dev = solvers._torch_device()
    name = None
    if torch is not None and dev is not None and dev.type == "cuda":
        try:
            name = torch.cuda.get_device_name(0)
        except Exception:                     # noqa: BLE001
            name = "cuda"
    return (backend, None if torch is None else dev.type, name, None if torch is None else torch.__version__)


def fingerprint(network, dt=0.001, seed=None, n_eval_points=None, solver_backend="auto", vco_shard=None, probes=None,
                prune=False, staged=True, neuron_shard=None, replicate=(), collapse=True):
    """Hex digest identifying what ``build(network, ...)`` would produce; raises ``Uncacheable`` when it cannot."""
    index, groups = _network_objects(network, probes)
    root = getattr(network, "seed", None)
    if root is None:
        root = seed
    if root is None:
        raise Uncacheable("neither the network nor the build call carries a seed")
    fp = _Fingerprint(index)
    fp.feed("sspslam_amd built model", FORMAT, _source_digest(), np.__version__)
    fp.value(dict(dt=float(dt), root_seed=int(root), n_eval_points=n_eval_points, solver=_resolved_solver(solver_backend),
                  vco_shard=vco_shard, neuron_shard=neuron_shard, prune=bool(prune), staged=bool(staged), collapse=bool(collapse)))
    fp.feed("replicate")
    for e in replicate:
        fp.ref(e)
    fp.feed("probes-override", probes is not None)
    for p in (probes or ()):
        fp.ref(p)
    for p in groups["x"]:
        _probe(fp, p)
    for w in groups["w"]:
        fp.feed("network", type(w).__qualname__, w.label, w.seed)
        fp.feed("members")
        for e in (getattr(w, "ea_ensembles", None) or ()):
            fp.ref(e)
    for e in groups["e"]:
        fp.feed("ensemble", e.n_neurons, e.dimensions, repr(e.radius), e.label, e.seed, bool(getattr(e, "normalize_encoders", True)))
        for attr in ("encoders", "intercepts", "max_rates", "eval_points", "n_eval_points", "neuron_type", "gain", "bias", "noise"):
            fp.feed(attr)
            v = getattr(e, attr, None)
            fp.feed("Default") if (v is fe.Default or type(v).__name__ == "DefaultType") else fp.value(v)
    for n in groups["n"]:
        fp.feed("node", n.size_in, n.size_out, n.label, n.seed)
        out = n.output
        if out is None:
            fp.feed("passthrough")
        elif callable(out):
            if n.size_in == 0:
                fp.feed("function of t")          # tabulated at run time from the live network: not part of the model
            else:
                fp.feed("function node")
                fp.value(getattr(n, "native", None))
                fp.value(out)
        else:
            fp.value(np.asarray(out, dtype=float))
    for c in groups["c"]:
        fp.feed("connection", c.label, c.seed, bool(c.scale_eval_points))
        fp.ref(c.pre)
        fp.ref(c.post)
        fp.value(None if c.synapse is None else c.synapse)
        fp.value(np.asarray(c.transform, dtype=float))
        fp.value(c.solver)
        fp.value(c.learning_rule_type)
        fp.value(c.eval_points)
        fp.feed("function")
        fp.value(c.function)
    for p in groups["p"]:
        _probe(fp, p)
    return fp.h.hexdigest()


def _probe(fp, p):
    fp.feed("probe", p.attr, p.label, p.seed, repr(p.sample_every))
    fp.ref(p.target)
    fp.value(p.synapse)
    fp.value(getattr(p, "unused", None))      # (sharding.py: decoded rows the probe's reader never looks at)


# ---- (de)serialisation: the model without the live objects ----------------------------------------------------------------
class _Pickler(pickle.Pickler):
    def __init__(self, f, index):
        super().__init__(f, protocol=5)
        self.index = index

    def persistent_id(self, obj):
        key = self.index.get(id(obj))
        if key is not None:
            return ("obj",) + key
        if isinstance(obj, fe.LearningRule):
            k = self.index.get(id(obj.connection))
            if k is not None:
                return ("rule",) + k
        if isinstance(obj, fe.Neurons):
            k = self.index.get(id(obj.ensemble))
            if k is not None:
                return ("neurons",) + k
        return None


class _Unpickler(pickle.Unpickler):
    def __init__(self, f, groups):
        super().__init__(f)
        self.groups = groups

    def persistent_load(self, pid):
        tag, kind, i = pid
        obj = self.groups[kind][i]
        if tag == "rule":
            return obj.learning_rule
        if tag == "neurons":
            return obj.neurons
        return obj


def _strip(model, index):
    """Shallow copy of the model with everything that is identity-keyed or live replaced by references."""
    m = BuiltModel.__new__(BuiltModel)
    m.__dict__.update(model.__dict__)
    m.tables = [{k: v for k, v in tb.items() if k != "fn"} for tb in model.tables]
    m.sig = {(name, index[key]) if key in index else (name, ("?", key)): v for (name, key), v in model.sig.items()}
    probes = []
    for p in model.probes:
        p = dict(p)
        b = p.get("buf")
        if isinstance(b, tuple) and len(b) == 2 and b[1] in index:
            p["buf"] = (b[0], ("@",) + index[b[1]])
        probes.append(p)
    m.probes = probes
    return m


def _rebind(m, groups):
    for tb in m.tables:
        tb["fn"] = tb["node"].output
    m.sig = {(name, id(groups[key[0]][key[1]]) if key[0] != "?" else key[1]): v for (name, key), v in m.sig.items()}
    for p in m.probes:
        b = p.get("buf")
        if isinstance(b, tuple) and len(b) == 2 and isinstance(b[1], tuple) and b[1][:1] == ("@",):
            p["buf"] = (b[0], id(groups[b[1][1]][b[1][2]]))
    return m


def save(model, network, path, extra_probes=None):
    index, _ = _network_objects(network, extra_probes)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = f"{path}.{os.getpid()}.tmp"
    try:
        with open(tmp, "wb") as f:          # (straight to the file: config 5's clean-up table alone is 14 GB)
            _Pickler(f, index).dump((FORMAT, _strip(model, index)))
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def load(path, network, extra_probes=None):
    _, groups = _network_objects(network, extra_probes)
    with open(path, "rb") as f:
        fmt, m = _Unpickler(f, groups).load()
    if fmt != FORMAT:
        raise Uncacheable("cache entry of another format")
    return _rebind(m, groups)


def cached_build(network, cache=None, **kw):
    """``builder.build(network, **kw)`` through the cache.  ``model.stats['cache']`` says what happened: "hit", "miss" (built and
    stored), "off" or "uncacheable: ..." ; ``model.stats['build_seconds']`` is the time THIS call took."""
    t0 = time.time()
    if cache is None:
        cache = os.environ.get("SSN_NO_CACHE", "") in ("", "0")
    if not cache:
        m = build(network, **kw)
        m.stats["cache"] = "off"
        return m
    try:
        key = fingerprint(network, **kw)
    except Uncacheable as e:
        m = build(network, **kw)
        m.stats["cache"] = f"uncacheable: {e}"
        return m
    path = os.path.join(cache_dir(), key + ".ssnmodel")
    if os.path.exists(path):
        try:
            m = load(path, network, kw.get("probes"))
            m.stats = dict(m.stats, cache="hit", cache_key=key, build_seconds_original=m.stats.get("build_seconds"),
                           build_seconds=time.time() - t0)
            return m
        except Exception as e:                # noqa: BLE001 - a damaged or stale entry is rebuilt, never trusted
            try:
                os.remove(path)
            except OSError:
                pass
            note = f"miss (entry unreadable: {e!r})"
    else:
        note = "miss"
    m = build(network, **kw)
    try:
        save(m, network, path, kw.get("probes"))
    except Exception as e:                    # noqa: BLE001 - a full disk must not fail the build
        note += f"; not stored: {e!r}"
    m.stats["cache"] = note
    m.stats["cache_key"] = key
    return m
