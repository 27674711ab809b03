"""nengo-shaped object model: just enough of nengo's front end to *declare* the SLAM networks.

The reference's networks are written against ``import nengo`` (call sites:
``networks/pathintegration.py:148-191``, ``networks/slam.py:241-307``, ``networks/binding.py:207-218,
292-324``, ``networks/associativememory.py:16-54``, ``experiments/run_pathint.py:75,120-148``).
nengo itself is a third-party package that is not part of the reference tree, so this module
provides objects with the same names, constructor arguments and attributes for the subset those
call sites use.  Objects only *describe* a model; ``builder.build`` turns them into arrays and
``simulator.Simulator`` runs them on the GPU.  ``Simulator`` traverses networks by attribute
(duck typing), so a genuine ``nengo.Network`` with the same attributes is accepted as well.

``install_as_nengo()`` registers this module tree as ``sys.modules['nengo']`` so that unmodified
model code (``import nengo; nengo.Network() ...``) runs on a machine without nengo.
"""
import sys
import types

import numpy as np

# --------------------------------------------------------------------------------------------
# sentinels / exceptions (nengo names)
# --------------------------------------------------------------------------------------------


class _DefaultType:
    def __repr__(self):
        return "Default"


Default = _DefaultType()


class NengoException(Exception):
    pass


class ValidationError(NengoException, ValueError):
    def __init__(self, msg, attr=None, obj=None):
        super().__init__(msg)
        self.attr, self.obj = attr, obj


class BuildError(NengoException, ValueError):
    pass


class SimulationError(NengoException, RuntimeError):
    pass


class ObsoleteError(NengoException):
    pass


class NetworkContextError(NengoException, RuntimeError):
    pass


# --------------------------------------------------------------------------------------------
# neuron types, synapses, solvers, learning rules
# --------------------------------------------------------------------------------------------
class NeuronType:
    spiking = False


class LIFRate(NeuronType):
    """Rate LIF: ``1/(tau_ref + tau_rc*log1p(1/(J-1)))`` for J>1."""

    def __init__(self, tau_rc=0.02, tau_ref=0.002, amplitude=1.0):
        self.tau_rc, self.tau_ref, self.amplitude = float(tau_rc), float(tau_ref), float(amplitude)

    def gain_bias(self, max_rates, intercepts):
        max_rates = np.asarray(max_rates, dtype=float)
        intercepts = np.asarray(intercepts, dtype=float)
        inv_tau_ref = 1.0 / self.tau_ref if self.tau_ref > 0 else np.inf
        if np.any(max_rates > inv_tau_ref):
            raise ValidationError("max_rates must be below 1/tau_ref", "max_rates", self)
        x = 1.0 / (1.0 - np.exp((self.tau_ref - 1.0 / max_rates) / self.tau_rc))
        gain = (1.0 - x) / (intercepts - 1.0)
        bias = 1.0 - gain * intercepts
        return gain, bias

    def rates(self, x, gain, bias):
        """Steady-state rates for decoded-space drive ``x`` (n_points, n_neurons)."""
        J = gain * x + bias
        out = np.zeros_like(J)
        m = J > 1
        out[m] = self.amplitude / (self.tau_ref + self.tau_rc * np.log1p(1.0 / (J[m] - 1.0)))
        return out


class LIF(LIFRate):
    """Spiking LIF."""
    spiking = True

    def __init__(self, tau_rc=0.02, tau_ref=0.002, min_voltage=0.0, amplitude=1.0):
        super().__init__(tau_rc, tau_ref, amplitude)
        self.min_voltage = float(min_voltage)


class RectifiedLinear(NeuronType):
    def __init__(self, amplitude=1.0):
        self.amplitude = float(amplitude)

    def gain_bias(self, max_rates, intercepts):
        max_rates = np.asarray(max_rates, dtype=float)
        intercepts = np.asarray(intercepts, dtype=float)
        gain = max_rates / (1.0 - intercepts)
        return gain, -intercepts * gain

    def rates(self, x, gain, bias):
        return self.amplitude * np.maximum(gain * x + bias, 0.0)


class Synapse:
    pass


class Lowpass(Synapse):
    """First-order low-pass; discretised as ``y <- a*y + (1-a)*u`` with ``a = exp(-dt/tau)`` (A.6)."""

    def __init__(self, tau):
        self.tau = float(tau)

    def __repr__(self):
        return f"Lowpass({self.tau})"


def _as_synapse(s, default_tau=0.005):
    if s is Default:
        return Lowpass(default_tau)
    if s is None or isinstance(s, Synapse):
        return s
    if np.isscalar(s):
        return Lowpass(float(s))
    if hasattr(s, "tau"):  # foreign (real nengo) Lowpass
        return Lowpass(float(s.tau))
    raise ValidationError(f"unsupported synapse {s!r}", "synapse")


class Solver:
    weights = False


class LstsqL2(Solver):
    """L2-regularised least squares, ``sigma = reg * max(A)``."""

    def __init__(self, weights=False, reg=0.1):
        self.weights, self.reg = bool(weights), float(reg)


class NoSolver(Solver):
    def __init__(self, values=None, weights=False):
        self.values, self.weights = values, bool(weights)


class LearningRuleType:
    pass


class PES(LearningRuleType):
    """Prescribed Error Sensitivity (A.7): ``W += -(lr*dt/n) * outer(error, filtered_activity)``."""

    def __init__(self, learning_rate=1e-4, pre_synapse=Default):
        self.learning_rate = float(learning_rate)
        self.pre_synapse = _as_synapse(pre_synapse)


class Voja(LearningRuleType):
    """Vector Oja encoder learning (A.8)."""

    def __init__(self, learning_rate=1e-2, post_synapse=Default):
        self.learning_rate = float(learning_rate)
        self.post_synapse = _as_synapse(post_synapse)


# --------------------------------------------------------------------------------------------
# distributions
# --------------------------------------------------------------------------------------------
class Distribution:
    def sample(self, n, d=None, rng=None):
        raise NotImplementedError


class Uniform(Distribution):
    def __init__(self, low, high):
        self.low, self.high = float(low), float(high)

    def sample(self, n, d=None, rng=None):
        rng = rng if rng is not None else np.random
        return rng.uniform(self.low, self.high, size=n if d is None else (n, d))


class Choice(Distribution):
    def __init__(self, options, weights=None):
        self.options, self.weights = np.asarray(options), weights

    def sample(self, n, d=None, rng=None):
        rng = rng if rng is not None else np.random
        idx = rng.choice(len(self.options), size=n, p=self.weights)
        return self.options[idx]


class UniformHypersphere(Distribution):
    """Uniform in the unit ball / on its surface: normalised Gaussian rows, radius ``u**(1/d)``."""

    def __init__(self, surface=False, min_magnitude=0.0):
        self.surface, self.min_magnitude = bool(surface), float(min_magnitude)

    def sample(self, n, d=None, rng=None):
        rng = rng if rng is not None else np.random
        d = 1 if d is None else d
        s = rng.randn(n, d)
        s /= np.linalg.norm(s, axis=1, keepdims=True)
        if not self.surface:
            lo = self.min_magnitude ** d
            s *= (lo + (1.0 - lo) * rng.rand(n, 1)) ** (1.0 / d)
        return s


class ScatteredHypersphere(UniformHypersphere):
    """Low-discrepancy points in the ball / on the sphere.

    nengo >= 3.1 builds these from an R_d sequence pushed through a spherical-coordinate map and a
    random rotation.  Reproducing nengo's exact stream is a non-goal: here the R_d
    sequence (same generator as the reference's ``Rd_sampling``, ``utils/utils.py:41-55``, with a
    random shift) is mapped through the inverse normal CDF to directions, the last coordinate gives
    the radius, and the cloud is randomly rotated - the same distribution and comparable evenness.
    """

    def sample(self, n, d=None, rng=None):
        from scipy.special import ndtri
        rng = rng if rng is not None else np.random
        d = 1 if d is None else d
        if d == 1 and self.surface:
            return np.where(rng.rand(n, 1) < 0.5, -1.0, 1.0)
        k = d + (0 if self.surface else 1)
        g = 2.0
        for _ in range(30):
            g = (1.0 + g) ** (1.0 / (k + 1))
        alpha = (1.0 / g) ** np.arange(1, k + 1) % 1.0
        u = (rng.rand(k)[None, :] + alpha[None, :] * np.arange(1, n + 1)[:, None]) % 1.0
        u = np.clip(u, 1e-12, 1 - 1e-12)
        v = ndtri(u[:, :d])
        v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-300)
        if d > 1:
            q, r = np.linalg.qr(rng.randn(d, d))
            v = v @ (q * np.sign(np.diag(r)))
        else:
            v = np.where(u[:, :1] < 0.5, -1.0, 1.0)
        if not self.surface:
            lo = self.min_magnitude ** d
            v = v * (lo + (1.0 - lo) * u[:, d:d + 1]) ** (1.0 / d)
        return v


class CosineSimilarity(Distribution):
    """Distribution of the cosine similarity between random unit vectors in ``dimensions``-D."""

    def __init__(self, dimensions):
        self.dimensions = int(dimensions)

    def sample(self, n, d=None, rng=None):
        rng = rng if rng is not None else np.random
        shape = (n,) if d is None else (n, d)
        g = rng.randn(*shape, self.dimensions)
        return g[..., 0] / np.linalg.norm(g, axis=-1)


# --------------------------------------------------------------------------------------------
# processes
# --------------------------------------------------------------------------------------------
class WhiteSignal:
    """Band-limited white noise; ``run(T, dt)`` -> (steps, 1) array."""

    def __init__(self, period, high, rms=0.5, y0=None, seed=None):
        self.period, self.high, self.rms, self.seed = float(period), float(high), float(rms), seed

    def run(self, t, dt=0.001, d=1, rng=None):
        rng = np.random.RandomState(self.seed)
        n_coef = int(np.ceil(self.period / dt / 2.0))
        sigma = self.rms * np.sqrt(0.5)
        out = np.empty((int(np.round(t / dt)), d))
        for j in range(d):
            coef = 1j * rng.normal(0.0, sigma, n_coef + 1)
            coef += rng.normal(0.0, sigma, n_coef + 1)
            coef[0] = 0.0
            coef[-1] = coef[-1].real
            freqs = np.fft.rfftfreq(2 * n_coef, d=dt)
            kill = freqs > self.high
            coef[kill] = 0.0
            coef /= np.sqrt(1.0 - kill.sum() / n_coef)
            coef *= np.sqrt(2 * n_coef)
            sig = np.fft.irfft(coef)
            out[:, j] = sig[np.arange(out.shape[0]) % sig.size]
        return out


# --------------------------------------------------------------------------------------------
# object model
# --------------------------------------------------------------------------------------------
class _ClassParams:
    """``model.config[nengo.Ensemble].neuron_type = ...`` (run_pathint.py:121)."""

    def __init__(self):
        self.__dict__["_v"] = {}

    def __getattr__(self, k):
        try:
            return self._v[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self._v[k] = v


class Config:
    def __init__(self):
        self._per_class = {}

    def __getitem__(self, cls):
        return self._per_class.setdefault(cls, _ClassParams())

    def default(self, cls, key, fallback):
        p = self._per_class.get(cls)
        return p._v.get(key, fallback) if p is not None else fallback


class Network:
    """Container with a context-manager build stack (nengo.Network)."""
    context = []

    def __init__(self, label=None, seed=None, add_to_container=None):
        self.label, self.seed = label, seed
        self.ensembles, self.nodes, self.connections, self.probes, self.networks = [], [], [], [], []
        self.config = Config()
        if Network.context and add_to_container is not False:
            Network.context[-1].networks.append(self)

    def __enter__(self):
        Network.context.append(self)
        return self

    def __exit__(self, *exc):
        if not Network.context or Network.context[-1] is not self:
            raise NetworkContextError("Network context stack corrupted")
        Network.context.pop()

    @staticmethod
    def add(obj):
        if not Network.context:
            raise NetworkContextError(f"{type(obj).__name__} must be created inside a `with Network():` block")
        net = Network.context[-1]
        for cls, lst in ((Ensemble, net.ensembles), (Node, net.nodes), (Connection, net.connections),
                         (Probe, net.probes)):
            if isinstance(obj, cls):
                lst.append(obj)
                return
        raise NetworkContextError(f"cannot add {obj!r}")

    def _all(self, attr):
        out = list(getattr(self, attr))
        for sub in self.networks:
            out.extend(sub._all(attr))
        return out

    all_ensembles = property(lambda self: self._all("ensembles"))
    all_nodes = property(lambda self: self._all("nodes"))
    all_connections = property(lambda self: self._all("connections"))
    all_probes = property(lambda self: self._all("probes"))

    @property
    def all_networks(self):
        out = list(self.networks)
        for sub in self.networks:
            out.extend(sub.all_networks)
        return out

    @property
    def n_neurons(self):
        return sum(e.n_neurons for e in self.all_ensembles)

    def __repr__(self):
        return f"<{type(self).__name__} {self.label!r}>"


def _effective_default(cls, key, fallback):
    for net in reversed(Network.context):
        v = net.config.default(cls, key, None)
        if v is not None:
            return v
    return fallback


class ObjView:
    """Slice of an object's input/output vector (``node[:d]``, ``ens.neurons[:k]``)."""

    def __init__(self, obj, key):
        self.obj = obj
        size = max(obj.size_in, obj.size_out)
        idx = np.arange(size)[key]
        self.indices = np.atleast_1d(idx).astype(np.int64)
        self.slice = key
        self.size_in = self.size_out = int(self.indices.size)

    def __repr__(self):
        return f"{self.obj!r}[{self.slice}]"


class _Sliceable:
    def __getitem__(self, key):
        return ObjView(self, key)


class Node(_Sliceable):
    """Passthrough (``output=None``), constant (array), or function node (A.2)."""

    def __init__(self, output=None, size_in=None, size_out=None, label=None, seed=None):
        self.label, self.seed = label, seed
        self.size_in = 0 if size_in is None else int(size_in)
        self._output = None
        self.size_out = self.size_in
        self.native = None  # optional ("identity"|"cleanup"|"gate", params): maps a function node to a kernel
        self._declared_size_out = size_out
        self.output = output
        Network.add(self)

    @property
    def output(self):
        return self._output

    @output.setter
    def output(self, value):
        self._output = value
        if value is None:
            self.size_out = self.size_in
        elif callable(value):
            if self._declared_size_out is not None:
                self.size_out = int(self._declared_size_out)
            else:
                args = (0.0,) if self.size_in == 0 else (0.0, np.zeros(self.size_in))
                try:
                    self.size_out = int(np.asarray(value(*args)).size)
                except Exception:
                    # closures indexing data with int((t-dt)/dt) at t=0 hit row -1: fine; anything
                    # else needs an explicit size_out
                    args = (0.001,) if self.size_in == 0 else (0.001, np.zeros(self.size_in))
                    self.size_out = int(np.asarray(value(*args)).size)
        else:
            value = np.asarray(value, dtype=float)
            self._output = value
            self.size_out = int(value.size)

    def __repr__(self):
        return f"<Node {self.label!r}>"


class Neurons(_Sliceable):
    def __init__(self, ensemble):
        self.ensemble = ensemble
        self.size_in = self.size_out = ensemble.n_neurons

    def __repr__(self):
        return f"<Neurons of {self.ensemble!r}>"


class Ensemble(_Sliceable):
    def __init__(self, n_neurons, dimensions, radius=1.0, encoders=Default, intercepts=Default,
                 max_rates=Default, eval_points=Default, n_eval_points=Default, neuron_type=Default,
                 gain=Default, bias=Default, noise=None, normalize_encoders=True, label=None, seed=None):
        if n_neurons <= 0 or dimensions <= 0:
            raise ValidationError("n_neurons and dimensions must be positive", "n_neurons", self)
        self.n_neurons, self.dimensions, self.radius = int(n_neurons), int(dimensions), float(radius)
        self.encoders, self.intercepts, self.max_rates = encoders, intercepts, max_rates
        self.eval_points, self.n_eval_points = eval_points, n_eval_points
        self.neuron_type = _effective_default(Ensemble, "neuron_type", LIF()) if neuron_type is Default \
            else neuron_type
        self.gain, self.bias, self.normalize_encoders = gain, bias, normalize_encoders
        self.label, self.seed = label, seed
        self.size_in = self.size_out = self.dimensions
        self.neurons = Neurons(self)
        Network.add(self)

    def __repr__(self):
        return f"<Ensemble {self.label!r} {self.n_neurons}x{self.dimensions}>"


class LearningRule:
    """Handle returned by ``conn.learning_rule``: a connection target (error / gate) and probeable."""

    def __init__(self, connection, learning_rule_type):
        self.connection, self.learning_rule_type = connection, learning_rule_type
        if isinstance(learning_rule_type, Voja):
            self.size_in = 1
        elif isinstance(learning_rule_type, PES):
            self.size_in = connection.size_out
        else:
            raise ValidationError(f"unsupported learning rule {learning_rule_type!r}", "learning_rule_type")
        self.size_out = 0

    def __repr__(self):
        return f"<LearningRule {type(self.learning_rule_type).__name__} on {self.connection!r}>"


class Connection:
    def __init__(self, pre, post, synapse=Default, function=None, transform=Default, solver=Default,
                 learning_rule_type=None, eval_points=None, scale_eval_points=True, label=None, seed=None):
        self.pre, self.post = pre, post
        self.synapse = _as_synapse(synapse)
        self.function = function
        self.transform = 1.0 if transform is Default else transform
        self.solver = LstsqL2() if solver is Default else solver
        self.learning_rule_type = learning_rule_type
        self.eval_points, self.scale_eval_points = eval_points, scale_eval_points
        self.label, self.seed = label, seed
        self.pre_obj = pre.obj if isinstance(pre, ObjView) else pre
        self.post_obj = post.obj if isinstance(post, ObjView) else post
        if function is not None and not isinstance(self.pre_obj, Ensemble):
            if not isinstance(self.pre_obj, Node):
                raise ValidationError("function only allowed on connections from an Ensemble or Node", "function")
        self.size_in = pre.size_out
        if function is None:
            self.size_mid = self.size_in
        else:
            self.size_mid = int(np.asarray(function(np.zeros(self.size_in))).size)
        self.size_out = post.size_in
        t = np.asarray(self.transform, dtype=float)
        if t.ndim == 2 and t.shape != (self.size_out, self.size_mid):
            raise ValidationError(f"transform shape {t.shape} != ({self.size_out}, {self.size_mid})", "transform")
        if t.ndim < 2 and self.size_out != self.size_mid:
            raise ValidationError(f"size mismatch {self.size_mid} -> {self.size_out} without a matrix transform",
                                  "transform")
        self.learning_rule = LearningRule(self, learning_rule_type) if learning_rule_type is not None else None
        Network.add(self)

    def __repr__(self):
        return f"<Connection {self.label or ''} {self.pre!r} -> {self.post!r}>"


class Probe:
    def __init__(self, target, attr=None, sample_every=None, synapse=None, label=None, seed=None):
        self.target, self.sample_every, self.label, self.seed = target, sample_every, label, seed
        self.obj = target.obj if isinstance(target, ObjView) else target
        if attr is None:
            if isinstance(self.obj, Connection):
                attr = "output"
            elif isinstance(self.obj, LearningRule):
                attr = "delta"
            else:
                attr = "output" if not isinstance(self.obj, Ensemble) else "decoded_output"
        self.attr = attr
        self.synapse = _as_synapse(synapse) if synapse is not None else None
        self.size_in = 0
        Network.add(self)

    def __repr__(self):
        return f"<Probe {self.attr} of {self.target!r}>"


class EnsembleArray(Network):
    """``nengo.networks.EnsembleArray``: ``n_ensembles`` equal ensembles."""

    def __init__(self, n_neurons, n_ensembles, ens_dimensions=1, label=None, seed=None,
                 add_to_container=None, **ens_kwargs):
        super().__init__(label=label, seed=seed, add_to_container=add_to_container)
        self.n_neurons_per_ensemble = int(n_neurons)
        self.n_ensembles = int(n_ensembles)
        self.dimensions_per_ensemble = int(ens_dimensions)
        self.ens_kwargs = ens_kwargs
        label_prefix = "" if label is None else label + "_"
        with self:
            self.input = Node(size_in=self.dimensions, label="input")
            self.ea_ensembles = []
            self._input_conns = []
            for i in range(self.n_ensembles):
                e = Ensemble(n_neurons, ens_dimensions, label=f"{label_prefix}{i}", **ens_kwargs)
                ed = ens_dimensions
                self._input_conns.append(Connection(self.input[i * ed:(i + 1) * ed], e, synapse=None))
                self.ea_ensembles.append(e)
        self.output = self.add_output("output", function=None)

    @property
    def dimensions(self):
        return self.n_ensembles * self.dimensions_per_ensemble

    def add_output(self, name, function, synapse=None, **conn_kwargs):
        ed = self.dimensions_per_ensemble
        if function is None:
            sizes = [ed] * self.n_ensembles
            funcs = [None] * self.n_ensembles
        elif callable(function):
            sz = int(np.asarray(function(np.zeros(ed))).size)
            sizes, funcs = [sz] * self.n_ensembles, [function] * self.n_ensembles
        else:
            funcs = list(function)
            sizes = [int(np.asarray(f(np.zeros(ed))).size) for f in funcs]
        with self:
            out = Node(size_in=int(np.sum(sizes)), label=name)
            off = 0
            for e, f, sz in zip(self.ea_ensembles, funcs, sizes):
                Connection(e, out[off:off + sz], function=f, synapse=synapse, **conn_kwargs)
                off += sz
        setattr(self, name, out)
        return out


# --------------------------------------------------------------------------------------------
# module tree with nengo's layout
# --------------------------------------------------------------------------------------------
def _ns(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


dists = _ns("nengo.dists", Distribution=Distribution, Uniform=Uniform, Choice=Choice,
            UniformHypersphere=UniformHypersphere, ScatteredHypersphere=ScatteredHypersphere,
            CosineSimilarity=CosineSimilarity)
solvers = _ns("nengo.solvers", Solver=Solver, LstsqL2=LstsqL2, NoSolver=NoSolver)
processes = _ns("nengo.processes", WhiteSignal=WhiteSignal)
synapses = _ns("nengo.synapses", Synapse=Synapse, Lowpass=Lowpass)
exceptions = _ns("nengo.exceptions", NengoException=NengoException, ValidationError=ValidationError,
                 BuildError=BuildError, SimulationError=SimulationError, ObsoleteError=ObsoleteError,
                 NetworkContextError=NetworkContextError)
network = _ns("nengo.network", Network=Network)
node = _ns("nengo.node", Node=Node)
connection = _ns("nengo.connection", Connection=Connection, LearningRule=LearningRule)
_ea_mod = _ns("nengo.networks.ensemblearray", EnsembleArray=EnsembleArray)
networks = _ns("nengo.networks", EnsembleArray=EnsembleArray, ensemblearray=_ea_mod)
_utils_numpy = _ns("nengo.utils.numpy", is_integer=lambda x: isinstance(x, (int, np.integer)),
                   is_number=lambda x: isinstance(x, (int, float, np.number)))
utils = _ns("nengo.utils", numpy=_utils_numpy)
rc = {"progress": {"progress_bar": None}}


def install_as_nengo(force=False):
    """Register this object model as the importable package ``nengo`` (no-op if real nengo exists)."""
    if "nengo" in sys.modules and not force and not getattr(sys.modules["nengo"], "_sspslam_amd_compat", False):
        return sys.modules["nengo"]
    from . import simulator as _sim  # late: simulator imports this module
    top = _ns("nengo", Default=Default, Network=Network, Node=Node, Ensemble=Ensemble,
              Connection=Connection, Probe=Probe, LIF=LIF, LIFRate=LIFRate, RectifiedLinear=RectifiedLinear,
              Lowpass=Lowpass, PES=PES, Voja=Voja, Simulator=_sim.Simulator, dists=dists, solvers=solvers,
              processes=processes, synapses=synapses, exceptions=exceptions, network=network, node=node,
              connection=connection, networks=networks, utils=utils, rc=rc, _sspslam_amd_compat=True)
    top.__path__ = []
    for m in (top, dists, solvers, processes, synapses, exceptions, network, node, connection, networks,
              _ea_mod, utils, _utils_numpy):
        sys.modules[m.__name__] = m
    return top
