#!/usr/bin/env python3
"""Generate golden vectors from the reference's own NumPy code.

Runs ONLY in the build container, where the reference is mounted at /root/reference.  The
reference's stepping arithmetic lives in the third-party package ``nengo`` (absent), but its
SSP algebra, Fourier-layout matrices, binding transforms, input-function factories and the
network *constructors* are plain NumPy / declarative Python.  They are loaded here file by file
with small stand-in modules for ``nengo`` (a recording fake of the object model: construction
only declares objects), and their outputs on fixed inputs are written to ``tests/golden/*.npz``
and ``topology_*.json``.  Nothing from the reference is copied: fixtures hold inputs and outputs.

    python tests/golden/make_golden.py            # rewrites the fixtures next to this file

The stand-in ``UniformHypersphere.sample`` restates nengo's published algorithm
(``rng.randn(n, d)`` rows normalised; interior points scaled by ``rng.rand(n,1)**(1/d)``).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------------------------
# recording fake of the nengo object model
# --------------------------------------------------------------------------------------------
class _Default:
    def __repr__(self):
        return "Default"


Default = _Default()
_ctx = []
_all = {"nodes": [], "ensembles": [], "connections": [], "probes": [], "networks": []}


class Network:
    def __init__(self, label=None, seed=None, **kw):
        self.label, self.seed = label, seed
        self.nodes, self.ensembles, self.connections, self.probes, self.networks = [], [], [], [], []
        self.config = {}
        if _ctx:
            _ctx[-1].networks.append(self)
        _all["networks"].append(self)

    def __enter__(self):
        _ctx.append(self)
        return self

    def __exit__(self, *a):
        _ctx.pop()


class _View:
    def __init__(self, obj, sl):
        self.obj, self.slice = obj, sl

    @property
    def size_out(self):
        return len(np.arange(self.obj.size_out)[self.slice].reshape(-1))

    size_in = size_out


class _Sliceable:
    def __getitem__(self, sl):
        return _View(self, sl)


class Node(_Sliceable):
    def __init__(self, output=None, size_in=None, size_out=None, label=None):
        self.output, self.label = output, label
        self.size_in = 0 if size_in is None else size_in
        if output is None:
            self.size_out = self.size_in
        elif callable(output):
            if size_out is not None:
                self.size_out = size_out
            else:
                args = (0.001,) if self.size_in == 0 else (0.001, np.zeros(self.size_in))
                self.size_out = int(np.asarray(output(*args)).size)
        else:
            self.size_out = int(np.asarray(output).size)
        _ctx[-1].nodes.append(self)
        _all["nodes"].append(self)


class _Neurons(_Sliceable):
    def __init__(self, ens):
        self.ensemble = ens
        self.size_in = self.size_out = ens.n_neurons


class Ensemble(_Sliceable):
    def __init__(self, n_neurons, dimensions, radius=1.0, encoders=Default, intercepts=Default,
                 max_rates=Default, eval_points=Default, neuron_type=Default, label=None, **kw):
        self.n_neurons, self.dimensions, self.radius = n_neurons, dimensions, radius
        self.encoders, self.intercepts, self.label = encoders, intercepts, label
        self.size_in = self.size_out = dimensions
        self.neurons = _Neurons(self)
        _ctx[-1].ensembles.append(self)
        _all["ensembles"].append(self)


class _LearningRule:
    def __init__(self, conn, rule_type):
        self.connection, self.learning_rule_type = conn, rule_type
        self.size_in = 1 if isinstance(rule_type, Voja) else conn.size_out


class Connection:
    def __init__(self, pre, post, synapse=Default, function=None, transform=Default,
                 solver=Default, learning_rule_type=None, eval_points=None, label=None, **kw):
        self.pre, self.post, self.synapse, self.function = pre, post, synapse, function
        self.transform, self.solver, self.learning_rule_type = transform, solver, learning_rule_type
        self.label = label
        self.size_out = post.size_in
        self.learning_rule = _LearningRule(self, learning_rule_type) if learning_rule_type else None
        _ctx[-1].connections.append(self)
        _all["connections"].append(self)


class Probe:
    def __init__(self, target, attr=None, synapse=None, sample_every=None, **kw):
        self.target, self.attr, self.synapse = target, attr, synapse
        _ctx[-1].probes.append(self)
        _all["probes"].append(self)


class PES:
    def __init__(self, learning_rate=1e-4, pre_synapse=Default):
        self.learning_rate = learning_rate


class Voja:
    def __init__(self, learning_rate=1e-2, post_synapse=Default):
        self.learning_rate, self.post_synapse = learning_rate, post_synapse


class LstsqL2:
    def __init__(self, weights=False, reg=0.1):
        self.weights, self.reg = weights, reg


class _Dist:
    def __init__(self, *a, **kw):
        self.args, self.kw = a, kw


class UniformHypersphere(_Dist):
    def __init__(self, surface=False, min_magnitude=0):
        self.surface = surface

    def sample(self, n, d=None, rng=np.random):
        s = rng.randn(n, d)
        s /= np.linalg.norm(s, axis=1, keepdims=True)
        if not self.surface:
            s *= rng.rand(n, 1) ** (1.0 / d)
        return s


class ScatteredHypersphere(UniformHypersphere):
    def __init__(self, surface=False, min_magnitude=0, **kw):
        self.surface = surface

    def sample(self, n, d=None, rng=None):
        return UniformHypersphere(self.surface).sample(n, d, rng=np.random.RandomState(12345))


class Choice(_Dist):
    pass


class EnsembleArray(Network):
    def __init__(self, n_neurons, n_ensembles, ens_dimensions=1, label=None, **ens_kwargs):
        super().__init__(label=label)
        self.n_ensembles, self.dimensions_per_ensemble = n_ensembles, ens_dimensions
        self.n_neurons_per_ensemble = n_neurons
        with self:
            self.input = Node(size_in=n_ensembles * ens_dimensions, label="input")
            self.ea_ensembles = []
            for i in range(n_ensembles):
                e = Ensemble(n_neurons, ens_dimensions, label=f"{label}_{i}", **ens_kwargs)
                Connection(self.input[i * ens_dimensions:(i + 1) * ens_dimensions], e, synapse=None)
                self.ea_ensembles.append(e)
        self.output = self.add_output("output", None)

    def add_output(self, name, function, synapse=None, **kw):
        if function is None:
            sz = self.dimensions_per_ensemble
        else:
            sz = int(np.asarray(function(np.zeros(self.dimensions_per_ensemble))).size)
        with self:
            out = Node(size_in=self.n_ensembles * sz, label=name)
            for i, e in enumerate(self.ea_ensembles):
                Connection(e, out[i * sz:(i + 1) * sz], function=function, synapse=synapse, **kw)
        setattr(self, name, out)
        return out


def _install_fake_nengo():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    dists = mod("nengo.dists", Distribution=_Dist, UniformHypersphere=UniformHypersphere,
                ScatteredHypersphere=ScatteredHypersphere, Choice=Choice, CosineSimilarity=_Dist)
    solvers = mod("nengo.solvers", LstsqL2=LstsqL2)
    network = mod("nengo.network", Network=Network)
    node = mod("nengo.node", Node=Node)
    connection = mod("nengo.connection", Connection=Connection)
    exceptions = mod("nengo.exceptions", ObsoleteError=RuntimeError, ValidationError=ValueError)
    ea = mod("nengo.networks.ensemblearray", EnsembleArray=EnsembleArray)
    networks = mod("nengo.networks", EnsembleArray=EnsembleArray, ensemblearray=ea)
    npx = mod("nengo.utils.numpy", is_integer=lambda x: isinstance(x, (int, np.integer)))
    utils = mod("nengo.utils", numpy=npx)
    mod("nengo", Network=Network, Node=Node, Ensemble=Ensemble, Connection=Connection, Probe=Probe,
        PES=PES, Voja=Voja, Default=Default, dists=dists, solvers=solvers, network=network,
        node=node, connection=connection, exceptions=exceptions, networks=networks, utils=utils)


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    _install_fake_nengo()
    # register empty packages by hand: sspslam/utils/__init__.py needs LaTeX + nengo_loihi
    for pkg in ("sspslam", "sspslam.utils", "sspslam.networks"):
        p = types.ModuleType(pkg)
        p.__path__ = []
        sys.modules[pkg] = p
    utils = _load("sspslam.utils.utils", "sspslam/utils/utils.py")
    sys.modules["sspslam.utils"].sparsity_to_x_intercept = utils.sparsity_to_x_intercept
    sys.modules["sspslam.utils"].Rd_sampling = utils.Rd_sampling
    ssp = _load("sspslam.sspspace", "sspslam/sspspace.py")
    pi = _load("sspslam.networks.pathintegration", "sspslam/networks/pathintegration.py")
    bind = _load("sspslam.networks.binding", "sspslam/networks/binding.py")
    am = _load("sspslam.networks.associativememory", "sspslam/networks/associativememory.py")
    nets = sys.modules["sspslam.networks"]
    nets.PathIntegration, nets.CircularConvolution = pi.PathIntegration, bind.CircularConvolution
    nets.Product, nets.AssociativeMemory = bind.Product, am.AssociativeMemory
    slam = _load("sspslam.networks.slam", "sspslam/networks/slam.py")
    slam_view = _load("sspslam.networks.slam_view", "sspslam/networks/slam_view.py")
    return dict(utils=utils, ssp=ssp, pi=pi, bind=bind, am=am, slam=slam, slam_view=slam_view)


# --------------------------------------------------------------------------------------------
def _syn(s):
    if s is Default:
        return "default"
    return None if s is None else float(s)


def _name(o):
    if isinstance(o, _View):
        sl = o.slice
        if isinstance(sl, slice):
            return f"{_name(o.obj)}[{sl.start}:{sl.stop}]"
        return f"{_name(o.obj)}[{sl}]"
    if isinstance(o, _Neurons):
        return f"{_name(o.ensemble)}.neurons"
    if isinstance(o, _LearningRule):
        return f"rule({type(o.learning_rule_type).__name__})"
    return f"{type(o).__name__}:{getattr(o, 'label', None)}"


def census(start):
    nodes = _all["nodes"][start["nodes"]:]
    ens = _all["ensembles"][start["ensembles"]:]
    conns = _all["connections"][start["connections"]:]
    nets = _all["networks"][start["networks"]:]
    syn_hist = {}
    n_func = 0
    tshapes = {}
    for c in conns:
        k = str(_syn(c.synapse))
        syn_hist[k] = syn_hist.get(k, 0) + 1
        n_func += c.function is not None
        if c.transform is not Default:
            shp = str(tuple(np.shape(c.transform)))
            tshapes[shp] = tshapes.get(shp, 0) + 1
    ens_hist = {}
    for e in ens:
        k = f"{e.n_neurons}x{e.dimensions}"
        ens_hist[k] = ens_hist.get(k, 0) + 1
    return dict(n_networks=len(nets), n_nodes=len(nodes), n_ensembles=len(ens),
                n_neurons=int(sum(e.n_neurons for e in ens)), n_connections=len(conns),
                synapse_hist=syn_hist, n_with_function=int(n_func), transform_shapes=tshapes,
                ensemble_hist=ens_hist,
                node_sizes=sorted([[n.label or "", int(n.size_in), int(n.size_out)] for n in nodes
                                   if n.label not in ("input", "output", "square")]),
                learning_rules=[type(c.learning_rule_type).__name__ for c in conns
                                if c.learning_rule_type is not None],
                first_connections=[[_name(c.pre), _name(c.post), _syn(c.synapse),
                                    c.function is not None] for c in conns[:12]],
                last_connections=[[_name(c.pre), _name(c.post), _syn(c.synapse),
                                   c.function is not None] for c in conns[-12:]])


def _mark():
    return {k: len(v) for k, v in _all.items()}


def synth_path(T, dt, dim, seed):
    """Deterministic smooth path in [-0.9, 0.9]^dim (fixture input, not nengo's WhiteSignal)."""
    t = np.arange(int(round(T / dt))) * dt
    rng = np.random.RandomState(seed)
    path = np.zeros((t.size, dim))
    for i in range(dim):
        f = rng.uniform(0.05, 0.4, size=4)
        ph = rng.uniform(0, 2 * np.pi, size=4)
        a = rng.uniform(0.3, 1.0, size=4)
        x = sum(a[j] * np.sin(2 * np.pi * f[j] * t + ph[j]) for j in range(4))
        path[:, i] = 1.8 * (x - x.min()) / (x.max() - x.min()) - 0.9
    return path


def main():
    R = load_reference()
    ssp, pi, bind, utils, slam = R["ssp"], R["pi"], R["bind"], R["utils"], R["slam"]
    rs = np.random.RandomState(2023)
    bounds2 = np.tile([-1.0, 1.0], (2, 1))

    # ---- G1-G4: spaces ---------------------------------------------------------------------
    g = {}
    pts16 = rs.uniform(-1, 1, size=(16, 2))
    g["pts16"] = pts16
    for d_req in (55, 97, 1015, 4033, 3000):
        s = ssp.HexagonalSSPSpace(2, ssp_dim=d_req, domain_bounds=bounds2, length_scale=0.2)
        d = s.ssp_dim
        g[f"hex2_req{d_req}_dim"] = np.array(d)
        A = s.phase_matrix
        if d <= 97:
            g[f"hex2_{d}_phase"] = A
        g[f"hex2_{d}_phase_sum"] = np.array([A.sum(), np.abs(A).sum(), (A ** 2).sum()])
        g[f"hex2_{d}_phase_rows"] = A[[1, 2, 3, d // 2, d // 2 + 1, d - 1]]
        g[f"hex2_{d}_rownorms"] = np.unique(np.round(np.linalg.norm(A, axis=1), 10))
        E = s.encode(pts16)
        g[f"hex2_{d}_enc16"] = E if d <= 1015 else E[:, :64]
    s24 = ssp.HexagonalSSPSpace(2, n_rotates=24, n_scales=28, domain_bounds=bounds2, length_scale=0.2)
    g["hex2_r24_s28_dim"] = np.array(s24.ssp_dim)
    g["hex2_r24_s28_phase_sum"] = np.array([s24.phase_matrix.sum(), np.abs(s24.phase_matrix).sum(),
                                            (s24.phase_matrix ** 2).sum()])
    s5x5 = ssp.HexagonalSSPSpace(2, domain_bounds=bounds2, length_scale=0.3)  # default 151
    g["hex2_default_dim"] = np.array(s5x5.ssp_dim)
    g["hex2_default_phase"] = s5x5.phase_matrix
    s1 = ssp.HexagonalSSPSpace(1, ssp_dim=37, domain_bounds=np.array([[-2.0, 2.0]]), length_scale=0.5)
    g["hex1_req37_dim"] = np.array(s1.ssp_dim)
    g["hex1_phase"] = s1.phase_matrix
    g["hex1_enc"] = s1.encode(np.linspace(-2, 2, 9)[:, None])
    s3 = ssp.HexagonalSSPSpace(3, ssp_dim=2047, domain_bounds=np.tile([-1.0, 1.0], (3, 1)),
                               length_scale=0.2, rng=np.random.default_rng(7))
    g["hex3_req2047_dim"] = np.array(s3.ssp_dim)
    g["hex3_phase"] = s3.phase_matrix
    pts3 = rs.uniform(-1, 1, size=(8, 3))
    g["hex3_pts"] = pts3
    g["hex3_enc_head"] = s3.encode(pts3)[:, :48]
    s3b = ssp.HexagonalSSPSpace(3, n_rotates=16, n_scales=16, domain_bounds=np.tile([-1.0, 1.0], (3, 1)),
                                length_scale=0.2, rng=np.random.default_rng(7))
    g["hex3_r16_s16_dim"] = np.array(s3b.ssp_dim)

    s55 = ssp.HexagonalSSPSpace(2, ssp_dim=55, domain_bounds=bounds2, length_scale=0.2)
    ss, sp = s55.get_sample_pts_and_ssps(100)
    g["hex2_55_grid_pts_head"] = sp[:205]
    g["hex2_55_grid_pts_sum"] = np.array([sp.sum(), (sp[:, 0] * np.arange(sp.shape[0])).sum()])
    g["hex2_55_grid_ssps_rows"] = ss[[0, 1, 99, 100, 5050, 9999]]
    g["hex2_55_grid_ssps_sum"] = np.array([ss.sum(), np.abs(ss).sum()])
    lsp = s55.get_sample_points(method="length-scale")
    g["hex2_55_ls_pts"] = lsp
    noisy = s55.encode(pts16) + 0.05 * rs.randn(16, 55)
    noisy[3] = 0.0
    noisy[4] *= 1e-8
    g["hex2_55_noisy"] = noisy
    g["hex2_55_decoded"] = s55.decode(noisy, "from-set", "grid", 100)
    g["hex2_55_decoded_31"] = s55.decode(noisy, "from-set", "grid", 31)
    s1015 = ssp.HexagonalSSPSpace(2, ssp_dim=1015, domain_bounds=bounds2, length_scale=0.2)
    ss, sp = s1015.get_sample_pts_and_ssps(100)
    g["hex2_1015_grid_ssps_sum"] = np.array([ss.sum(), np.abs(ss).sum()])
    g["hex2_1015_grid_row5050_head"] = ss[5050, :64]
    # G7 algebra
    a, b = rs.randn(3, 55), rs.randn(3, 55)
    g["alg_a"], g["alg_b"] = a, b
    g["alg_bind"] = s55.bind(a, b)
    g["alg_invert"] = s55.invert(a)
    g["alg_unitary"] = np.stack([s55.make_unitary(a[i]) for i in range(3)])
    g["alg_normalize"] = s55.normalize(a[0])
    g["alg_identity"] = s55.identity()
    # RandomSSPSpace shape rule only (rng-dependent content)
    r = ssp.RandomSSPSpace(2, ssp_dim=64, domain_bounds=bounds2, rng=np.random.default_rng(3))
    g["rand_req64_dim"] = np.array(r.ssp_dim)
    g["rand_req64_phase"] = r.phase_matrix
    # G15 SPSpace
    for (n, d, seed) in ((10, 55, 0), (10, 1015, 0), (5, 97, 3), (1, 9, 0)):
        sps = ssp.SPSpace(n, d, seed=seed)
        v = sps.vectors
        g[f"sp_{n}_{d}_{seed}_vectors"] = v if d <= 97 else v[:, :32]
        g[f"sp_{n}_{d}_{seed}_gram"] = v @ v.T
        g[f"sp_{n}_{d}_{seed}_inv0"] = sps.inverse_vectors[0][:32]
    sps = ssp.SPSpace(10, 55, seed=0)
    g["sp_bind01"] = sps.bind(sps.vectors[0], sps.vectors[1])
    g["sp_decode"] = sps.decode(sps.vectors[[3, 1, 7]] + 0.01)
    g["sp_bindmat"] = sps.get_binding_matrix(sps.vectors[2:3])
    # sample_grid_encoders (sobol path needs scipy qmc with the space rng -> 'grid' only)
    s55.rng = np.random.default_rng(11)
    g["hex2_55_gridenc"] = s55.sample_grid_encoders(40, method="grid")
    np.savez_compressed(os.path.join(OUT, "ssp_spaces.npz"), **g)

    # ---- G5/G6: Fourier layout + binding transforms -------------------------------------------
    g = {}
    for d in (7, 8, 55):
        g[f"to_fourier_{d}"] = pi.get_to_Fourier(d)
        g[f"from_fourier_{d}"] = pi.get_from_Fourier(d)
        for al in "AB":
            for inv in (False, True):
                g[f"tr_in_{d}_{al}_{int(inv)}"] = bind.transform_in(d, al, inv)
        g[f"tr_out_{d}"] = bind.transform_out(d)
    tf, ff = pi.get_to_Fourier(1015), pi.get_from_Fourier(1015)
    g["to_fourier_1015_sum"] = np.array([tf.sum(), np.abs(tf).sum(), (tf * tf).sum()])
    g["from_fourier_1015_sum"] = np.array([ff.sum(), np.abs(ff).sum(), (ff * ff).sum()])
    g["to_fourier_1015_rows"] = tf[[3, 4, 5, 760, 1522], :40]
    g["from_fourier_1015_rows"] = ff[[0, 1, 507, 1014], :40]
    ta, to = bind.transform_in(1015, "A", False), bind.transform_out(1015)
    g["tr_in_1015_A_sum"] = np.array([ta.sum(), np.abs(ta).sum()])
    g["tr_out_1015_sum"] = np.array([to.sum(), np.abs(to).sum()])
    x = rs.randn(8, 55)
    y = rs.randn(8, 55)
    g["cc_x"], g["cc_y"] = x, y
    g["cc_xy"] = bind.circconv(x, y)
    g["cc_xy_inva"] = bind.circconv(x, y, invert_a=True)
    g["cc_xy_invb"] = bind.circconv(x, y, invert_b=True)
    g["dft_half_7"] = np.stack([bind.dft_half(7).real, bind.dft_half(7).imag])
    np.savez_compressed(os.path.join(OUT, "fourier_binding.npz"), **g)

    # ---- G8: utils ------------------------------------------------------------------------
    g = {}
    g["rd_10_2_0"] = utils.Rd_sampling(10, 2, 0)
    g["rd_20_3_0"] = utils.Rd_sampling(20, 3, 0)
    g["rd_7_2_default"] = utils.Rd_sampling(7, 2)
    g["sparsity_in"] = np.array([[55, 0.1], [1015, 0.1], [97, 0.5], [55, 0.8], [3, 0.25]])
    g["sparsity_out"] = np.array([utils.sparsity_to_x_intercept(int(d), p) for d, p in g["sparsity_in"]])
    np.savez_compressed(os.path.join(OUT, "utils.npz"), **g)

    # ---- G10 + G11: PathIntegration constructor (feedback closure, transforms, topology) -------
    g = {}
    topo = {}
    m0 = _mark()
    with Network(seed=0) as model:
        net = pi.PathIntegration(s55, 500, 0.05, scaling_factor=0.3, stable=True, solver_weights=False)
    topo["pi_d55_n500"] = census(m0)
    fb = net.recur_conns[0].function
    grid = np.stack(np.meshgrid(np.linspace(-1.2, 1.2, 7), np.linspace(-1.2, 1.2, 7),
                                np.linspace(-1, 1, 5), indexing="ij"), -1).reshape(-1, 3)
    grid = np.vstack([grid, [[0.0, 0.0, 0.3], [0.6, 0.2, 0.5]]])
    g["fb_grid"] = grid
    g["fb_stable_tau05_sf03"] = np.stack([fb(p) for p in grid])
    g["pi_vel_transforms"] = np.stack([c.transform for c in net.connections
                                       if isinstance(c.transform, np.ndarray) and c.transform.shape == (3, 2)])
    g["pi_phase_matrix"] = s55.phase_matrix
    with Network(seed=0):
        net2 = pi.PathIntegration(s55, 100, 0.1, scaling_factor=1.0, stable=True, max_radius=0.8)
        net3 = pi.PathIntegration(s55, 100, 0.05, scaling_factor=0.5, stable=False)
    g["fb_stable_tau1_sf1_r08"] = np.stack([net2.recur_conns[3].function(p) for p in grid])
    g["fb_sho_tau05_sf05"] = np.stack([net3.recur_conns[3].function(p) for p in grid])
    np.savez_compressed(os.path.join(OUT, "pathintegration.npz"), **g)

    # ---- G9: SLAM input functions on a synthetic path + SLAM topology ---------------------------
    g = {}
    T, dt = 2.0, 0.001
    path = synth_path(T, dt, 2, seed=5)
    vels = (1 / dt) * np.diff(path, axis=0, prepend=path[0:1])
    obj_locs = 0.9 * 2 * (utils.Rd_sampling(10, 2, seed=0) - 0.5)
    vec_to_lm = obj_locs[None, :, :] - path[:, None, :]
    lm_space = ssp.SPSpace(10, 55, seed=0)
    g["path"], g["vels"], g["obj_locs"] = path, vels, obj_locs
    n_steps = path.shape[0]
    ts = np.arange(1, n_steps + 1) * dt
    for tag, fn in (("f1", slam.get_slam_input_functions), ("f2", slam.get_slam_input_functions2)):
        vf, scale, inview, idf, spf, vecf, vecsspf = fn(s55, lm_space, vels, vec_to_lm, 0.2)
        g[f"{tag}_scale"] = np.array(scale)
        g[f"{tag}_vel"] = np.stack([vf(t) for t in ts])
        g[f"{tag}_inview"] = np.array([inview(t) for t in ts])
        g[f"{tag}_sp"] = np.stack([spf(t) for t in ts]).astype(np.float32)
        g[f"{tag}_vecssp"] = np.stack([vecsspf(t) for t in ts]).astype(np.float32)
        g[f"{tag}_vec"] = np.stack([vecf(t) for t in ts])
    g["idx_t_minus_dt"] = np.array([int((t - dt) / dt) for t in np.arange(1, 200001) * dt], dtype=np.int32)
    g["idx_floor_t"] = np.array([int(np.minimum(np.floor(t / dt), 200000 - 2))
                                 for t in np.arange(1, 200001) * dt], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "slam_inputs.npz"), **g)

    m0 = _mark()
    with Network(seed=0):
        sl = slam.SLAMNetwork(s55, lm_space, 0.2, 10, 500, 550, 100, tau_pi=0.05,
                              update_thres=0.2, vel_scaling_factor=0.3, shift_rate=0.2,
                              voja_learning_rate=1e-4, pes_learning_rate=5e-3, clean_up_method="grid",
                              gc_n_neurons=0, encoders=None, voja=True, seed=0, intercept=0.1)
    topo["slam_d55_pi500_m550_c100_lm10"] = census(m0)
    g = {}
    xg = rs.randn(6, 2 * 55 + 1) * 0.2
    xg[0, -1] = 0.0
    xg[0, :55] = s55.encode(np.array([[0.1, 0.2]]))[0]
    xg[0, 55:110] = s55.encode(np.array([[0.12, 0.22]]))[0]
    xg[1] = xg[0]
    xg[1, -1] = 10.0
    xg[2] = xg[0]
    xg[2, -1] = 0.0009
    xg[3] = xg[0]
    xg[3, 55:110] = s55.encode(np.array([[-0.7, 0.6]]))[0]
    xg[4, -1] = 0.0
    xg[5, -1] = -0.002
    g["gate_x"] = xg
    g["gate_out"] = np.stack([sl.update_state.output(0.1, x) for x in xg])
    xc = s55.encode(rs.uniform(-1, 1, (5, 2))) + 0.1 * rs.randn(5, 55)
    g["cleanup_x"] = xc
    g["cleanup_out"] = np.stack([sl.clean_up_fun(x) for x in xc])
    unit_fn = [c.function for c in sl.connections if c.function is not None][0]
    g["unitary_x"] = xc
    g["unitary_out"] = np.stack([unit_fn(x) for x in xc])
    np.savez_compressed(os.path.join(OUT, "slam_nodes.npz"), **g)

    # SLAMViewNetwork (slam_view.py): topology + input tables on the same synthetic path
    sv = R["slam_view"]
    m0 = _mark()
    with Network(seed=0):
        sv.SLAMViewNetwork(s55, lm_space, 0.2, 10, 500, 550, 100, tau_pi=0.05, update_thres=0.2,
                           vel_scaling_factor=0.3, shift_rate=0.2, voja_learning_rate=1e-4,
                           pes_learning_rate=5e-3, clean_up_method="grid", gc_n_neurons=0, encoders=None,
                           voja=True, seed=0)
    topo["slamview_d55_pi500_m550_lm10"] = census(m0)
    g = {}
    vf, scale, inview, lmf = sv.get_slamview_input_functions(s55, lm_space, vels, vec_to_lm, 0.2)
    ts_v = ts[:1500]
    g["scale"] = np.array(scale)
    g["vel"] = np.stack([vf(t) for t in ts_v])
    g["inview"] = np.array([inview(t) for t in ts_v])
    g["view_ssp"] = np.stack([lmf(t) for t in ts_v]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "slamview_inputs.npz"), **g)

    m0 = _mark()
    with Network(seed=0):
        bind.CircularConvolution(100, 55, invert_a=True, label="cc")
    topo["circconv_d55_c100_inva"] = census(m0)
    with open(os.path.join(OUT, "topology.json"), "w") as f:
        json.dump(topo, f, indent=1, sort_keys=True)
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn:28s} {os.path.getsize(os.path.join(OUT, fn)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: fixtures can only be regenerated in the build container")
    main()
