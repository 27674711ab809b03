"""SSP-SLAM from egocentric *views*: the landmark input is already the bound SSP ``lm (*) S(vector)``.

Mirrors the reference's ``SLAMViewNetwork`` (``sspslam/networks/slam_view.py:7-276``) and
``get_slamview_input_functions`` (``:281-404``): the SLAMNetwork graph without the two circular-convolution
networks - the view SSP keys the associative memory directly, the memory's value is the cleaned-up
path-integrator estimate, and its (unitary-projected) recall drives the loop-closure gate.  Function nodes
with inputs are tagged ``node.native`` exactly as in ``slam.py``.
"""
import numpy as np

from .. import frontend as nengo
from .associativememory import AssociativeMemory
from .pathintegration import PathIntegration
from .slam import make_cleanup, make_gate, _unitary_fn


class SLAMViewNetwork(nengo.Network):
    def __init__(self, ssp_space, lm_space, view_rad, n_landmarks, pi_n_neurons, mem_n_neurons,
                 circonv_n_neurons, tau=0.01, tau_pi=0.05, update_thres=0.2, vel_scaling_factor=1.0,
                 rad_scaling_factor=1.0, shift_rate=0.1, voja_learning_rate=5e-4, pes_learning_rate=1e-2,
                 clean_up_method="grid", gc_n_neurons=0, encoders=None, voja=True, seed=0):
        super().__init__()
        if clean_up_method != "grid":
            raise nengo.BuildError("only clean_up_method='grid' is on the hot path (SURVEY §2 row 6)")
        domain_dim, d = ssp_space.domain_dim, ssp_space.ssp_dim
        rng = np.random.RandomState(seed)
        landmark_sps = lm_space.vectors
        if (not voja) and encoders is None:
            encoders = landmark_sps[rng.randint(n_landmarks, size=mem_n_neurons)]
        intercept = (landmark_sps @ landmark_sps.T - np.eye(n_landmarks)).max()      # no 0.5 cap here (slam_view.py:200)
        self.sample_ssps, self.sample_points = ssp_space.get_sample_pts_and_ssps(100)
        self.clean_up_fun = make_cleanup(self.sample_ssps)
        self.grid_factors = ssp_space.grid_factors(100) if hasattr(ssp_space, "grid_factors") else None
        cleanup = self.clean_up_fun

        with self:
            self.velocity_input = nengo.Node(size_in=domain_dim, label="vel_input")
            self.view_input = nengo.Node(size_in=d, label="lm_input")
            self.no_landmark_in_view = nengo.Node(size_in=1, label="lm_in_view_input")

            self.update_state = nengo.Node(make_gate(d, update_thres, shift_rate), size_in=2 * d + 1, size_out=d)
            self.update_state.native = ("gate", d, float(update_thres), float(shift_rate))
            nengo.Connection(self.no_landmark_in_view, self.update_state[-1], synapse=None)

            self.pathintegrator = PathIntegration(ssp_space, pi_n_neurons, tau_pi, max_radius=rad_scaling_factor,
                                                  scaling_factor=vel_scaling_factor, stable=True, label="pathint")
            self.output = self.pathintegrator.output
            nengo.Connection(self.velocity_input, self.pathintegrator.velocity_input, synapse=None)
            nengo.Connection(self.update_state, self.pathintegrator.input, synapse=None)

            self.assomemory = AssociativeMemory(mem_n_neurons, d, d, intercept,
                                                voja_learning_rate=voja_learning_rate,
                                                pes_learning_rate=pes_learning_rate, voja=voja, encoders=encoders)
            nengo.Connection(self.view_input, self.assomemory.key_input, synapse=None)
            nengo.Connection(self.no_landmark_in_view, self.assomemory.learning, synapse=None)

            if gc_n_neurons <= 0:
                self.gridcells = nengo.Node(lambda t, x: cleanup(x), size_in=d, size_out=d)
                self.gridcells.native = ("cleanup", self.sample_ssps, self.grid_factors)
                nengo.Connection(self.pathintegrator.output, self.gridcells, synapse=tau)
                nengo.Connection(self.gridcells, self.assomemory.value_input, synapse=None)
            else:
                self.cleanup = nengo.Node(lambda t, x: cleanup(x), size_in=d, size_out=d)
                self.cleanup.native = ("cleanup", self.sample_ssps, self.grid_factors)
                self.gridcells = nengo.Ensemble(gc_n_neurons, d, encoders=ssp_space.sample_grid_encoders(gc_n_neurons),
                                                intercepts=nengo.CosineSimilarity(d + 2))
                nengo.Connection(self.pathintegrator.output, self.cleanup, synapse=tau)
                nengo.Connection(self.cleanup, self.gridcells, synapse=None)
                nengo.Connection(self.gridcells, self.assomemory.value_input, synapse=tau)

            nengo.Connection(self.assomemory.recall, self.update_state[:d], function=_unitary_fn(ssp_space), synapse=tau)
            nengo.Connection(self.pathintegrator.output, self.update_state[d:-1], synapse=tau)


def get_slamview_input_functions(ssp_space, lm_space, velocity_data, vec_to_landmarks_data, view_rad, dt=0.001):
    """Recorded data -> input-node functions (reference ``slam_view.py:281-404``).

    Returns ``(velocity_func, vel_scaling_factor, is_landmark_in_view, landmark_func)``:
    velocity scaled by ``1 / max|A v|``; flag 1 when no landmark is within ``view_rad`` else 0; the view SSP
    ``normalise(sum_i lm_i (*) S(vec_i))`` over the landmarks in view (zeros when none).  Row lookups follow
    the reference literally: ids and velocity use ``min(floor(t/dt), pathlen-2)``, the encoded vector uses
    ``int((t-dt)/dt)``."""
    velocity_data = np.asarray(velocity_data, dtype=float)
    vecs = np.asarray(vec_to_landmarks_data, dtype=float)          # (pathlen, n_landmarks, dim)
    pathlen = vecs.shape[0]
    d = ssp_space.ssp_dim
    sps = lm_space.vectors
    scale = 1.0 / np.max(np.abs(ssp_space.phase_matrix @ velocity_data.T))
    vels_scaled = velocity_data * scale

    def row_ahead(t):
        return int(np.minimum(np.floor(t / dt), pathlen - 2))

    def velocity_func(t):
        return vels_scaled[row_ahead(t)]

    def ids_in_view(t):
        return np.where(np.linalg.norm(vecs[row_ahead(t)], axis=1) < view_rad)[0]

    def landmark_func(t):
        out = np.zeros(d)
        r = int((t - dt) / dt)
        for i in ids_in_view(t):
            out += np.ravel(ssp_space.bind(sps[i], np.ravel(ssp_space.encode(vecs[r, i]))))
        nrm = np.linalg.norm(out)
        return out / nrm if nrm > 1e-8 else out

    def is_landmark_in_view(t):
        return 1 if len(ids_in_view(t)) == 0 else 0

    return velocity_func, scale, is_landmark_in_view, landmark_func
