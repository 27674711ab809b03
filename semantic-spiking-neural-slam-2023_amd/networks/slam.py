"""SSP-SLAM network and the helpers that turn recorded data into input-node functions.

Mirrors the reference's ``SLAMNetwork`` (``sspslam/networks/slam.py:8-307``) and
``get_slam_input_functions`` / ``get_slam_input_functions2`` (``:312-438`` / ``:442-497``): same
constructor arguments, node/ensemble attributes and connection topology (checked against
tests/golden/topology.json).  The two function nodes with inputs are tagged with ``node.native`` so
the simulator runs them as kernels instead of calling Python every step:

* ``gridcells``    ``("cleanup", sample_ssps)``  - ``S[argmax_j <S_j, x>]``        (``:212-215,270``)
* ``update_state`` ``("gate", d, thres, rate)``  - loop-closure gate                 (``:233-237,249``)
"""
import numpy as np

from .. import frontend as nengo
from .associativememory import AssociativeMemory
from .binding import CircularConvolution
from .pathintegration import PathIntegration


def make_cleanup(sample_ssps):
    def clean_up_fun(x):
        return sample_ssps[np.argmax(sample_ssps @ x)]
    return clean_up_fun


def make_gate(d, update_thres, shift_rate):
    """``x = [estimate(d), current(d), flag]`` -> ``rate*(estimate-current)`` when a landmark is in
    view (|flag| <= 1e-3, the ``np.allclose(flag, 0, atol=1e-3)`` of the reference) and the two SSPs
    agree better than ``update_thres``; zeros otherwise."""
    def update_state_func(t, x):
        est, cur, flag = x[:d], x[d:2 * d], x[2 * d]
        if abs(flag) <= 1e-3 and float(np.dot(est, cur)) > update_thres:
            return shift_rate * (est - cur)
        return np.zeros(d)
    return update_state_func


def _unitary_fn(ssp_space):
    def unitary(x):
        return ssp_space.make_unitary(x)

    def batch(X):
        F = np.fft.fft(np.asarray(X, dtype=float), axis=1)
        return np.fft.ifft(F / np.maximum(np.abs(F), 1e-8), axis=1).real
    unitary.batch = batch
    return unitary


class SLAMNetwork(nengo.Network):
    def __init__(self, ssp_space, lm_space, view_rad, n_landmarks, pi_n_neurons, mem_n_neurons,
                 circonv_n_neurons, tau=0.01, tau_pi=0.05, update_thres=0.2, vel_scaling_factor=1.0,
                 rad_scaling_factor=1.0, shift_rate=0.1, voja_learning_rate=5e-4, pes_learning_rate=1e-2,
                 clean_up_method="grid", gc_n_neurons=0, encoders=None, voja=True, seed=0,
                 landmark_sps=None, intercept=None):
        super().__init__()
        if clean_up_method != "grid":
            raise nengo.BuildError("only clean_up_method='grid' is on the hot path (SURVEY §2 row 6)")
        domain_dim, d = ssp_space.domain_dim, ssp_space.ssp_dim
        rng = np.random.RandomState(seed)
        if landmark_sps is None:
            landmark_sps = lm_space.vectors
        if (not voja) and encoders is None:
            encoders = landmark_sps[rng.randint(n_landmarks, size=mem_n_neurons)]
        if intercept is None:
            off_diag = landmark_sps @ landmark_sps.T - np.eye(n_landmarks)
            intercept = min(off_diag.max(), 0.5)

        ovc_pts = nengo.ScatteredHypersphere(surface=False, min_magnitude=1e-3).sample(
            mem_n_neurons, domain_dim, rng=np.random.RandomState(seed + 1))
        ovc_encoders = ssp_space.encode(ovc_pts)
        self.sample_ssps, self.sample_points = ssp_space.get_sample_pts_and_ssps(100)
        self.clean_up_fun = make_cleanup(self.sample_ssps)
        if hasattr(ssp_space, "grid_factors"):
            self.grid_factors = ssp_space.grid_factors(100)
        else:                    # (a space object of the reference's own class: the factors are recovered from the table)
            from ..sspspace import grid_factors_from_table
            self.grid_factors = grid_factors_from_table(self.sample_ssps)
        unitary = _unitary_fn(ssp_space)

        with self:
            self.velocity_input = nengo.Node(size_in=domain_dim, label="vel_input")
            self.landmark_id_input = nengo.Node(size_in=d, label="lm_id_input")
            self.landmark_vec_ssp = nengo.Node(size_in=d, label="lm_vecssp_input")
            self.no_landmark_in_view = nengo.Node(size_in=1, label="lm_in_view_input")

            self.update_state = nengo.Node(make_gate(d, update_thres, shift_rate), size_in=2 * d + 1,
                                           size_out=d)
            self.update_state.native = ("gate", d, float(update_thres), float(shift_rate))
            nengo.Connection(self.no_landmark_in_view, self.update_state[-1], synapse=None)

            self.pathintegrator = PathIntegration(ssp_space, pi_n_neurons, tau_pi,
                                                  max_radius=rad_scaling_factor,
                                                  scaling_factor=vel_scaling_factor, stable=True,
                                                  solver_weights=False, label="pathint")
            self.output = self.pathintegrator.output
            nengo.Connection(self.velocity_input, self.pathintegrator.velocity_input, synapse=None)
            nengo.Connection(self.update_state, self.pathintegrator.input, synapse=None)

            self.ovc_ens = nengo.Ensemble(mem_n_neurons, d, encoders=ovc_encoders)
            nengo.Connection(self.landmark_vec_ssp, self.ovc_ens, synapse=None)
            self.landmark_ssp_ens = CircularConvolution(circonv_n_neurons, dimensions=d,
                                                        label="landmark_circonv")
            nengo.Connection(self.ovc_ens, self.landmark_ssp_ens.input_b, synapse=None)

            cleanup = self.clean_up_fun
            if gc_n_neurons <= 0:
                self.gridcells = nengo.Node(lambda t, x: cleanup(x), size_in=d, size_out=d)
                self.gridcells.native = ("cleanup", self.sample_ssps, self.grid_factors)
                nengo.Connection(self.pathintegrator.output, self.gridcells, synapse=tau)
                nengo.Connection(self.gridcells, self.landmark_ssp_ens.input_a, synapse=None)
            else:
                self.cleanup = nengo.Node(lambda t, x: cleanup(x), size_in=d, size_out=d)
                self.cleanup.native = ("cleanup", self.sample_ssps, self.grid_factors)
                self.gridcells = nengo.Ensemble(gc_n_neurons, d,
                                                encoders=ssp_space.sample_grid_encoders(gc_n_neurons),
                                                intercepts=nengo.CosineSimilarity(d + 2))
                nengo.Connection(self.pathintegrator.output, self.cleanup, synapse=tau)
                nengo.Connection(self.cleanup, self.gridcells, synapse=None)
                nengo.Connection(self.gridcells, self.landmark_ssp_ens.input_a, synapse=tau)

            self.assomemory = AssociativeMemory(mem_n_neurons, d, d, intercept,
                                                voja_learning_rate=voja_learning_rate,
                                                pes_learning_rate=pes_learning_rate, voja=voja,
                                                encoders=encoders)
            nengo.Connection(self.landmark_id_input, self.assomemory.key_input, synapse=None)
            nengo.Connection(self.landmark_ssp_ens.output, self.assomemory.value_input, synapse=tau)
            nengo.Connection(self.no_landmark_in_view, self.assomemory.learning, synapse=None)

            self.position_estimate = CircularConvolution(circonv_n_neurons, d, invert_a=True,
                                                         label="newpos_circonv")
            nengo.Connection(self.ovc_ens, self.position_estimate.input_a, synapse=tau, function=unitary)
            nengo.Connection(self.assomemory.recall, self.position_estimate.input_b, synapse=tau,
                             function=unitary)
            nengo.Connection(self.position_estimate.output, self.update_state[:d], synapse=tau)
            nengo.Connection(self.pathintegrator.output, self.update_state[d:-1], synapse=tau)


# --------------------------------------------------------------------------------------------
# input-function factories
# --------------------------------------------------------------------------------------------
def _make_input_functions(ssp_space, lm_space, velocity_data, vec_to_landmarks_data, view_rad, dt, multi):
    velocity_data = np.asarray(velocity_data, dtype=float)
    vecs = np.asarray(vec_to_landmarks_data, dtype=float)          # (pathlen, n_landmarks, dim)
    pathlen, _, domain_dim = vecs.shape
    d = ssp_space.ssp_dim
    sps = lm_space.vectors
    scale = 1.0 / np.max(np.abs(ssp_space.phase_matrix @ velocity_data.T))
    vels_scaled = velocity_data * scale

    def row(t):
        # the reference's float64 expression, evaluated literally (it is NOT step-1)
        return int((t - dt) / dt)

    def row_ahead(t):
        return int(np.minimum(np.floor(t / dt), pathlen - 2))

    def velocity_func(t):
        return vels_scaled[row(t)]

    def landmark_id_func(t):
        dist = np.linalg.norm(vecs[row(t)], axis=1)
        if np.all(dist > view_rad):
            return None if multi else -1
        return np.where(dist <= view_rad)[0] if multi else int(np.argmin(dist))

    def _ids(t):
        cur = landmark_id_func(t)
        if multi:
            return None if cur is None else list(cur)
        return None if cur < 0 else [cur]

    def landmark_vec_func(t):
        ids = _ids(t)
        if ids is None:
            return np.zeros(domain_dim)
        return np.sum([vecs[row(t), i] for i in ids], axis=0)

    def landmark_sp_func(t):
        ids = _ids(t)
        if ids is None:
            return np.zeros(d)
        return np.sum([sps[i] for i in ids], axis=0)

    def landmark_vecssp_func(t):
        ids = _ids(t)
        if ids is None:
            return np.zeros(d)
        r = row_ahead(t)
        return np.sum(ssp_space.encode(vecs[r, ids]), axis=0)

    def is_landmark_in_view(t):
        return 10 if _ids(t) is None else 0

    return (velocity_func, scale, is_landmark_in_view, landmark_id_func, landmark_sp_func,
            landmark_vec_func, landmark_vecssp_func)


def get_slam_input_functions(ssp_space, lm_space, velocity_data, vec_to_landmarks_data, view_rad, dt=0.001):
    """Closest-landmark inputs (reference ``slam.py:312-438``).  Returns the same 7-tuple."""
    return _make_input_functions(ssp_space, lm_space, velocity_data, vec_to_landmarks_data, view_rad, dt, False)


def get_slam_input_functions2(ssp_space, lm_space, velocity_data, vec_to_landmarks_data, view_rad, dt=0.001):
    """All-landmarks-in-view inputs, summed (reference ``slam.py:442-497``)."""
    return _make_input_functions(ssp_space, lm_space, velocity_data, vec_to_landmarks_data, view_rad, dt, True)
