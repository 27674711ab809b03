"""Experiment harness: synthetic paths, model assembly and metrics of the reference's run scripts.

Re-creates, as functions, what ``experiments/run_pathint.py`` and ``experiments/run_slam.py`` do
around ``sim.run`` so that tests, ``bench.py`` and the CLI share
one definition of the benchmark workloads:

* ``make_random_path``      ``run_pathint.py:72-89``   band-limited white-noise path rescaled to +-0.9
* ``make_pathint_model``    ``run_pathint.py:105-143`` input nodes + PathIntegration + probe
* ``pathint_metrics``       ``run_pathint.py:168-184`` grid decode, cosine similarity, distance error
* ``make_slam_model``       ``run_slam.py:95-195``
* ``map_recall``            ``run_slam.py:263-268``
"""
import numpy as np

from . import frontend as nengo
from .networks import PathIntegration, SLAMNetwork, get_slam_input_functions2
from .sspspace import HexagonalSSPSpace, SPSpace
from .utils import Rd_sampling


def make_random_path(T, dt=0.001, domain_dim=2, limit=0.1, seed=0, radius=1.0):
    """(path, vels): per-axis ``WhiteSignal(T, high=limit, seed=seed+axis)`` min-max scaled to
    +-0.9*radius; ``vels = diff(path)/dt`` with a leading zero row."""
    path = np.hstack([nengo.WhiteSignal(T, high=limit, seed=seed + i).run(T, dt=dt) for i in range(domain_dim)])
    for i in range(domain_dim):
        x = path[:, i]
        path[:, i] = 1.8 * radius * (x - x.min()) / (x.max() - x.min()) - 0.9 * radius
    vels = np.diff(path, axis=0, prepend=path[:1]) / dt
    return path, vels


def make_ssp_space(domain_dim=2, ssp_dim=97, n_scales=0, n_rotates=3, length_scale=0.2, radius=1.0, rng=None):
    bounds = radius * np.tile([-1.0, 1.0], (domain_dim, 1))
    if n_scales > 0:
        return HexagonalSSPSpace(domain_dim, n_scales=n_scales, n_rotates=n_rotates, domain_bounds=bounds,
                                 length_scale=length_scale, rng=rng)
    return HexagonalSSPSpace(domain_dim, ssp_dim=ssp_dim, domain_bounds=bounds, length_scale=length_scale, rng=rng)


class PathIntModel:
    pass


def make_pathint_model(ssp_space, path, vels, pi_n_neurons, tau=0.05, neuron_type=None, seed=0, dt=0.001,
                       init_time=0.05, probe_synapse=0.05):
    """The model of ``run_pathint.py:105-143``; returns an object with ``model``, ``probe``,
    ``pathintegrator``, ``real_ssp``, ``scale_fac``."""
    d = ssp_space.ssp_dim
    real_ssp = ssp_space.encode(path)
    scale_fac = 1.0 / np.max(np.abs(ssp_space.phase_matrix @ vels.T))
    vels_scaled = vels * scale_fac
    out = PathIntModel()
    model = nengo.Network(seed=seed)
    if neuron_type is not None:
        model.config[nengo.Ensemble].neuron_type = neuron_type
    with model:
        vel_input = nengo.Node(lambda t: vels_scaled[int((t - dt) / dt)], label="vel_input")
        init_state = nengo.Node(lambda t: real_ssp[int((t - dt) / dt)] if t < init_time else np.zeros(d),
                                label="init_state")
        out.pathintegrator = PathIntegration(ssp_space, pi_n_neurons, tau, scaling_factor=scale_fac,
                                             stable=True, solver_weights=False)
        nengo.Connection(vel_input, out.pathintegrator.velocity_input, synapse=None)
        nengo.Connection(init_state, out.pathintegrator.input, synapse=None)
        out.probe = nengo.Probe(out.pathintegrator.output, synapse=probe_synapse)
    out.model, out.real_ssp, out.scale_fac, out.ssp_space = model, real_ssp, scale_fac, ssp_space
    out.path, out.vels = path, vels
    return out


def pathint_metrics(ssp_space, sim_out, real_ssp, path, num_samples=None):
    """(sim_path_est, pi_sims, pi_error) as ``run_pathint.py:179-184``."""
    if num_samples is None:
        num_samples = 100 if ssp_space.domain_dim < 3 else 50
    n = sim_out.shape[0]
    est = ssp_space.decode(sim_out, "from-set", "grid", num_samples)
    nrm = np.linalg.norm(sim_out, axis=1)
    sims = np.sum(sim_out * real_ssp[:n], axis=1) / np.where(nrm == 0, 1.0, nrm)
    err = np.sqrt(np.sum((path[:n] - est) ** 2, axis=1))
    return est, sims, err


def cosine_error(a, b):
    """Per-row ``1 - <a,b>/(|a||b|)`` between two probe trajectories (the parity observable)."""
    num = np.sum(a * b, axis=1)
    den = np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1)
    return 1.0 - num / np.where(den == 0, 1.0, den)


class SlamModel:
    pass


def make_slam_model(ssp_space, path, vels, n_landmarks=10, pi_n_neurons=500, mem_n_neurons=None,
                    circonv_n_neurons=100, view_rad=0.2, update_thres=0.2, shift_rate=0.2,
                    voja_learning_rate=1e-4, pes_learning_rate=5e-3, intercept=0.1, tau_pi=0.05, seed=0,
                    dt=0.001, init_time=0.05, weights_sample_every=None):
    """The model of ``run_slam.py:95-195`` (multi-landmark inputs, ``get_slam_input_functions2``)."""
    d = ssp_space.ssp_dim
    domain_dim = ssp_space.domain_dim
    if mem_n_neurons is None:
        mem_n_neurons = 10 * d
    obj_locs = 0.9 * 2 * (Rd_sampling(n_landmarks, domain_dim, seed=seed) - 0.5)
    vec_to_landmarks = obj_locs[None, :, :] - path[:, None, :]
    lm_space = SPSpace(n_landmarks, d, seed=seed)
    real_ssp = ssp_space.encode(path)
    (velocity_func, vel_scaling_factor, is_landmark_in_view, landmark_id_func, landmark_sp_func,
     landmark_vec_func, landmark_vecssp_func) = get_slam_input_functions2(
        ssp_space, lm_space, vels, vec_to_landmarks, view_rad, dt=dt)
    out = SlamModel()
    model = nengo.Network(seed=seed)
    with model:
        vel_input = nengo.Node(velocity_func, label="vel_input")
        init_state = nengo.Node(lambda t: real_ssp[int((t - dt) / dt)] if t < init_time else np.zeros(d),
                                label="init_state")
        landmark_vec = nengo.Node(landmark_vecssp_func, label="lm_vec")
        landmark_id = nengo.Node(landmark_sp_func, label="lm_id")
        is_landmark = nengo.Node(is_landmark_in_view, label="lm_in_view")
        out.slam = SLAMNetwork(ssp_space, lm_space, view_rad, n_landmarks, pi_n_neurons, mem_n_neurons,
                               circonv_n_neurons, tau_pi=tau_pi, update_thres=update_thres,
                               vel_scaling_factor=vel_scaling_factor, shift_rate=shift_rate,
                               voja_learning_rate=voja_learning_rate, pes_learning_rate=pes_learning_rate,
                               clean_up_method="grid", gc_n_neurons=0, encoders=None, voja=True, seed=seed,
                               intercept=intercept)
        nengo.Connection(vel_input, out.slam.velocity_input, synapse=None)
        nengo.Connection(init_state, out.slam.pathintegrator.input, synapse=None)
        nengo.Connection(landmark_vec, out.slam.landmark_vec_ssp, synapse=None)
        nengo.Connection(landmark_id, out.slam.landmark_id_input, synapse=None)
        nengo.Connection(is_landmark, out.slam.no_landmark_in_view, synapse=None)
        out.probe = nengo.Probe(out.slam.pathintegrator.output, synapse=0.05)
        out.weights_probe = nengo.Probe(out.slam.assomemory.conn_out, "weights",
                                        sample_every=weights_sample_every) if weights_sample_every else None
    out.model, out.real_ssp, out.ssp_space, out.lm_space = model, real_ssp, ssp_space, lm_space
    out.obj_locs, out.path, out.vels, out.vel_scaling_factor = obj_locs, path, vels, vel_scaling_factor
    return out


def get_activities(built_ens, neuron_type, x):
    """Rates of an ensemble for decoded-space points ``x`` (nengo.builder.ensemble.get_activities)."""
    proj = x @ built_ens.encoders.T / built_ens.radius
    return neuron_type.rates(proj, built_ens.gain, built_ens.bias)


def map_recall(ssp_space, lm_space, built_memory, neuron_type, final_decoders):
    """Definition (i) of map recall, ``run_slam.py:263-268``: activities of the *built* (pre-Voja)
    memory ensemble for each landmark SP times the final PES decoders -> decoded landmark positions."""
    acts = get_activities(built_memory, neuron_type, lm_space.vectors)
    recalled = acts @ np.asarray(final_decoders).T
    return recalled, ssp_space.decode(recalled, "from-set", "grid", 100)
