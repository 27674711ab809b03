"""Experiment harness: synthetic paths, model assembly and metrics of the reference's run scripts.

Re-creates, as functions, what ``experiments/run_pathint.py`` and ``experiments/run_slam.py`` do
around ``sim.run`` so that tests, ``bench.py`` and the CLI share
one definition of the benchmark workloads:

* ``make_random_path``      ``run_pathint.py:72-89``   band-limited white-noise path rescaled to +-0.9
* ``make_pathint_model``    ``run_pathint.py:105-143`` input nodes + PathIntegration + probe
* ``pathint_metrics``       ``run_pathint.py:168-184`` grid decode, cosine similarity, distance error
* ``make_slam_model``       ``run_slam.py:95-195``
* ``map_recall``            ``run_slam.py:263-268``        (definition (i): built encoders)
* ``map_recall_learned``    ``slam_map_new.py:342-347,447-452``  (definition (ii): Voja-probed encoders)
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
    if not np.all(np.isfinite(path)):
        raise nengo.ValidationError(f"a {T} s band-limited path has no frequency below {limit} Hz (need T >= {1.0 / limit} s)", "limit")
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


def indexed_rows_node_fn(table, dt, until=None):
    """The input-node closure of the reference scripts, ``lambda t: table[int((t - dt) / dt)]`` (``run_pathint.py:134``;
    zeros from ``t >= until`` on, ``:136``), plus a vectorised twin ``fn.table(steps) -> (rows, idx)`` that the HIP
    simulator's tabulation uses instead of one Python call per timestep (same float64 arithmetic, so the same rows)."""
    table = np.asarray(table, dtype=float)
    zero = np.zeros(table.shape[1])

    def fn(t):
        if until is not None and not t < until:
            return zero
        return table[int((t - dt) / dt)]

    def rows_for(steps):
        t = np.asarray(steps, dtype=np.int64) * dt                    # nengo's t = step * dt in float64
        k = ((t - dt) / dt).astype(np.int64)                          # int(): truncation
        live = np.ones(k.shape, dtype=bool) if until is None else t < until
        if not live.any():
            return np.zeros((0, table.shape[1])), np.full(k.shape, -1, dtype=np.int32)
        lo, hi = int(k[live].min()), int(k[live].max())
        idx = np.where(live, k - lo, -1).astype(np.int32)             # -1 = a row of zeros
        return table[lo:hi + 1], idx

    fn.table = rows_for
    return fn


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
        vel_input = nengo.Node(indexed_rows_node_fn(vels_scaled, dt), label="vel_input")
        init_state = nengo.Node(indexed_rows_node_fn(real_ssp, dt, until=init_time), label="init_state")
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
                    dt=0.001, init_time=0.05, weights_sample_every=None, obj_locs=None):
    """The model of ``run_slam.py:95-195`` (multi-landmark inputs, ``get_slam_input_functions2``).  ``obj_locs``: landmark
    positions (default: the script's ``0.9 * 2 * (Rd_sampling - 0.5)``, ``run_slam.py:119``) - tests place a landmark next to the
    start of the path so that PES and Voja learn from the first timesteps."""
    d = ssp_space.ssp_dim
    domain_dim = ssp_space.domain_dim
    if mem_n_neurons is None:
        mem_n_neurons = 10 * d
    if obj_locs is None:
        obj_locs = 0.9 * 2 * (Rd_sampling(n_landmarks, domain_dim, seed=seed) - 0.5)
    obj_locs = np.asarray(obj_locs, dtype=float).reshape(n_landmarks, domain_dim)
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


def make_config3_model(seed=0, T=20.0, dt=0.001, landmark_near_start=True, pi_n_neurons=10000, mem_n_neurons=10150,
                       circonv_n_neurons=100, n_landmarks=10, ssp_dim=1015, view_rad=0.2, weights_sample_every=None):
    """BASELINE configs[2]: SLAMNetwork 2-D, ssp_dim 1015, 10 000 neurons per VCO, 10 150 memory neurons, 10 landmarks
    (reference ``experiments/run_slam.py:24-31,180-185``), on the synthetic path of SURVEY 8d - ONE definition for
    ``bench.py``'s SLAM leg and the full-size GPU parity test.  ``landmark_near_start``: landmark 0 is moved to 0.07 from the
    path's first point (inside ``view_rad`` = 0.2), so that the landmark inputs, PES and Voja are live from the first
    timesteps - with the script's ``Rd_sampling`` positions the first landmark comes into view after ~200 timesteps, past any
    window the NumPy oracle can follow at this size."""
    space = make_ssp_space(2, ssp_dim)
    path, vels = make_random_path(T, dt=dt, limit=0.1, seed=seed)
    obj = 0.9 * 2 * (Rd_sampling(n_landmarks, 2, seed=seed) - 0.5)
    if landmark_near_start:
        obj[0] = path[0] + np.array([0.05, 0.05])
    return make_slam_model(space, path, vels, n_landmarks=n_landmarks, pi_n_neurons=pi_n_neurons, mem_n_neurons=mem_n_neurons,
                           circonv_n_neurons=circonv_n_neurons, view_rad=view_rad, seed=seed, dt=dt,
                           weights_sample_every=weights_sample_every, obj_locs=obj)


def make_slamview_model(ssp_space, path, vels, n_landmarks=10, pi_n_neurons=500, mem_n_neurons=None,
                        circonv_n_neurons=100, view_rad=0.2, update_thres=0.2, shift_rate=0.2,
                        voja_learning_rate=5e-4, pes_learning_rate=1e-3, tau_pi=0.05, seed=0, dt=0.001,
                        init_time=0.05, weights_sample_every=None):
    """The model of ``run_slamview.py:88-137``: SLAMViewNetwork fed with the bound view SSP of the landmarks in
    sight (``get_slamview_input_functions``)."""
    from .networks import SLAMViewNetwork, get_slamview_input_functions
    d = ssp_space.ssp_dim
    domain_dim = ssp_space.domain_dim
    if mem_n_neurons is None:
        mem_n_neurons = 10 * d
    obj_locs = 0.9 * 2 * (Rd_sampling(n_landmarks, domain_dim, seed=seed) - 0.5)
    vec_to_landmarks = obj_locs[None, :, :] - path[:, None, :]
    lm_space = SPSpace(n_landmarks, d, seed=seed)
    real_ssp = ssp_space.encode(path)
    velocity_func, vel_scaling_factor, is_landmark_in_view, landmark_func = get_slamview_input_functions(
        ssp_space, lm_space, vels, vec_to_landmarks, view_rad, dt=dt)
    out = SlamModel()
    model = nengo.Network(seed=seed)
    with model:
        vel_input = nengo.Node(velocity_func, label="vel_input")
        init_state = nengo.Node(lambda t: real_ssp[int((t - dt) / dt)] if t < init_time else np.zeros(d),
                                label="init_state")
        landmark_input = nengo.Node(landmark_func, label="lm_view")
        landmark_inview = nengo.Node(is_landmark_in_view, label="lm_in_view")
        out.slam = SLAMViewNetwork(ssp_space, lm_space, view_rad, n_landmarks, pi_n_neurons, mem_n_neurons,
                                   circonv_n_neurons, tau_pi=tau_pi, update_thres=update_thres,
                                   vel_scaling_factor=vel_scaling_factor, shift_rate=shift_rate,
                                   voja_learning_rate=voja_learning_rate, pes_learning_rate=pes_learning_rate,
                                   clean_up_method="grid", gc_n_neurons=0, encoders=None, voja=True, seed=seed)
        nengo.Connection(landmark_input, out.slam.view_input, synapse=None)
        nengo.Connection(landmark_inview, out.slam.no_landmark_in_view, synapse=None)
        nengo.Connection(vel_input, out.slam.velocity_input, synapse=None)
        nengo.Connection(init_state, out.slam.pathintegrator.input, synapse=None)
        out.probe = nengo.Probe(out.slam.pathintegrator.output, synapse=0.05)
        out.recall_probe = nengo.Probe(out.slam.assomemory.recall, synapse=0.05)
        out.weights_probe = nengo.Probe(out.slam.assomemory.conn_out, "weights",
                                        sample_every=weights_sample_every) if weights_sample_every else None
    out.model, out.real_ssp, out.ssp_space, out.lm_space = model, real_ssp, ssp_space, lm_space
    out.obj_locs, out.path, out.vels, out.vel_scaling_factor = obj_locs, path, vels, vel_scaling_factor
    return out


# --------------------------------------------------------------------------------------------
# recorded trajectories in, result files out (the reference scripts' formats)
# --------------------------------------------------------------------------------------------
def stretch_trajectory(traj, original_dt=0.02, new_dt=0.001):
    """Resample a recorded 2-D trajectory from ``original_dt`` to ``new_dt`` by linear interpolation
    (``run_pathint.py:57-66``)."""
    traj = np.asarray(traj, dtype=float)
    n = traj.shape[0]
    total = n * original_dt
    m = int(total / new_dt)
    t_old, t_new = np.linspace(0, total, n), np.linspace(0, total, m)
    return np.stack([np.interp(t_new, t_old, traj[:, 0]), np.interp(t_new, t_old, traj[:, 1])], axis=1)


def load_path(path_data, data_dt=0.001, dt=0.001, radius=1.0, max_rows=49999):
    """``--path-data`` of the reference scripts (``run_pathint.py:77-89``): load an ``.npy`` trajectory, cut it
    to ``max_rows``, resample if it was recorded at another step, rescale each axis to ±0.9·radius and
    differentiate.  Returns ``(path, vels)``."""
    path = np.load(path_data)[:max_rows, :].astype(float)
    if data_dt != dt:
        path = stretch_trajectory(path, original_dt=data_dt, new_dt=dt)
    for i in range(path.shape[1]):
        lo, hi = path[:, i].min(), path[:, i].max()
        path[:, i] = 1.8 * radius * (path[:, i] - lo) / (hi - lo) - 0.9 * radius
    vels = (1 / dt) * np.diff(path, axis=0, prepend=path[0:1])
    return path, vels


def save_pathint_results(filename, ssp_space, ts, path, real_ssp, pi_sim_out, elapsed_time, args=None,
                         elapsed_thread_time=None):
    """``np.savez`` with the field names of ``run_pathint.py:201-207`` so the reference's plotting scripts
    (``plot_trials_2d.py``) read our runs.  Returns the decoded path estimate."""
    pi_path, pi_sims, pi_error = pathint_metrics(ssp_space, pi_sim_out, real_ssp, path)
    n = pi_sim_out.shape[0]
    path, real_ssp = path[:n], real_ssp[:n]
    np.savez(filename, ts=ts, path=path, real_ssp=real_ssp, pi_sim_out=pi_sim_out, pi_sims=pi_sims, pi_path=pi_path,
             pi_error=pi_error, elapsed_time=elapsed_time,
             elapsed_thread_time=elapsed_time if elapsed_thread_time is None else elapsed_thread_time,
             args=np.array(repr(args)), sig_to_noise_ratio=np.array(np.nan))
    return pi_path


def save_slam_results(filename, ssp_space, ts, path, real_ssp, obj_locs, view_rad, slam_sim_out, landmark_ssps_est,
                      landmark_loc_est, elapsed_time, args=None, elapsed_thread_time=None):
    """Field names of ``run_slam.py:282-293``.  Returns the decoded path estimate."""
    slam_path, slam_sims, slam_error = pathint_metrics(ssp_space, slam_sim_out, real_ssp, path)
    n = slam_sim_out.shape[0]
    path, real_ssp = path[:n], real_ssp[:n]
    np.savez(filename, timesteps=np.arange(path.shape[0]) * (ts[1] - ts[0] if len(ts) > 1 else 0.001), ts=ts, path=path,
             real_ssp=real_ssp, obj_locs=obj_locs, view_rad=view_rad, slam_sim_out=slam_sim_out, slam_sims=slam_sims,
             slam_path=slam_path, slam_error=slam_error, landmark_ssps_est=landmark_ssps_est,
             landmark_loc_est=landmark_loc_est, elapsed_time=elapsed_time,
             elapsed_thread_time=elapsed_time if elapsed_thread_time is None else elapsed_thread_time,
             args=np.array(repr(args)), sig_to_noise_ratio=np.array(np.nan))
    return slam_path


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


def map_recall_learned(ssp_space, lm_space, built_memory, neuron_type, final_scaled_encoders, final_decoders,
                       num_samples=100):
    """Definition (ii) of map recall, ``slam_map_new.py:342-347`` / ``:447-452``: the landmark SPs are projected on the
    FINAL scaled encoders (``Probe(conn_in.learning_rule, "scaled_encoders")[-1]``, i.e. after Voja moved them) and
    pushed through ``neuron_type.rates(x, gain, bias)``, then through the final PES decoders.

    Reproduced as written, including its quirk: ``scaled_encoders`` already contain
    ``gain / radius`` and ``rates`` multiplies by ``gain`` once more."""
    x = np.asarray(lm_space.vectors) @ np.asarray(final_scaled_encoders).T
    acts = neuron_type.rates(x, built_memory.gain, built_memory.bias)
    recalled = acts @ np.asarray(final_decoders).T
    return recalled, ssp_space.decode(recalled, "from-set", "grid", num_samples)
