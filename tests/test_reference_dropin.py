"""Drop-in check at the source level: the reference's OWN network classes, loaded unchanged from /root/reference with
this repo's object model registered as ``import nengo`` (``frontend.install_as_nengo``), build through our builder and
step to the same trajectory as our own builders of the same networks.

Runs only where the reference checkout exists (the build container); it never travels to the GPU box, and nothing of
it is copied: the files are executed from where they lie."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "sspslam")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    import sspslam_amd.frontend as fe
    saved = {k: v for k, v in sys.modules.items() if k == "nengo" or k.startswith("nengo.") or k == "sspslam" or k.startswith("sspslam.")}
    fe.install_as_nengo(force=True)

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod
    # register the packages by hand: sspslam/utils/__init__.py wants LaTeX fonts and nengo_loihi
    for pkg in ("sspslam", "sspslam.utils", "sspslam.networks"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    utils = load("sspslam.utils.utils", "sspslam/utils/utils.py")
    sys.modules["sspslam.utils"].sparsity_to_x_intercept = utils.sparsity_to_x_intercept
    sys.modules["sspslam.utils"].Rd_sampling = utils.Rd_sampling
    out = types.SimpleNamespace(utils=utils)
    out.ssp = load("sspslam.sspspace", "sspslam/sspspace.py")
    out.pi = load("sspslam.networks.pathintegration", "sspslam/networks/pathintegration.py")
    out.bind = load("sspslam.networks.binding", "sspslam/networks/binding.py")
    out.am = load("sspslam.networks.associativememory", "sspslam/networks/associativememory.py")
    nets = sys.modules["sspslam.networks"]
    nets.PathIntegration, nets.CircularConvolution = out.pi.PathIntegration, out.bind.CircularConvolution
    nets.Product, nets.AssociativeMemory = out.bind.Product, out.am.AssociativeMemory
    out.slam = load("sspslam.networks.slam", "sspslam/networks/slam.py")
    import nengo
    out.nengo = nengo
    yield out
    for k in [k for k in sys.modules if k == "nengo" or k.startswith("nengo.") or k == "sspslam" or k.startswith("sspslam.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_reference_pathintegration_class_on_our_stack(ref):
    from sspslam_amd import harness as H
    from sspslam_amd.builder import build
    from oracle import OracleSimulator
    nengo, dt = ref.nengo, 0.001
    space_ref = ref.ssp.HexagonalSSPSpace(2, ssp_dim=55, domain_bounds=np.tile([-1.0, 1.0], (2, 1)), length_scale=0.2)
    space = H.make_ssp_space(2, 55)
    np.testing.assert_allclose(space_ref.phase_matrix, space.phase_matrix, atol=1e-14, rtol=0)
    path, vels = H.make_random_path(10.0, limit=0.2, seed=0)
    ours = H.make_pathint_model(space, path, vels, 60)
    real_ssp = space_ref.encode(path)
    scale_fac = 1.0 / np.max(np.abs(space_ref.phase_matrix @ vels.T))
    vs = vels * scale_fac
    model = nengo.Network(seed=0)
    with model:                                             # run_pathint.py:128-143, with the reference's class
        vel_input = nengo.Node(lambda t: vs[int((t - dt) / dt)], label="vel_input")
        init_state = nengo.Node(lambda t: real_ssp[int((t - dt) / dt)] if t < 0.05 else np.zeros(space_ref.ssp_dim))
        pint = ref.pi.PathIntegration(space_ref, 60, 0.05, scaling_factor=scale_fac, stable=True, solver_weights=False)
        nengo.Connection(vel_input, pint.velocity_input, synapse=None)
        nengo.Connection(init_state, pint.input, synapse=None)
        nengo.Probe(pint.output, synapse=0.05)
    bm_ref, bm = build(model), build(ours.model)
    assert bm_ref.stats["n_ops"] == bm.stats["n_ops"] and bm_ref.stats["n_neurons"] == bm.stats["n_neurons"] == 28 * 60
    a, b = OracleSimulator(bm_ref), OracleSimulator(bm)
    a.run_steps(150)
    b.run_steps(150)
    assert np.abs(b.probe_data(0)).max() > 0.1
    np.testing.assert_allclose(a.probe_data(0), b.probe_data(0), atol=1e-9, rtol=0)


def test_reference_slamnetwork_class_on_our_stack(ref):
    """The reference's SLAMNetwork (with its CircularConvolution, Product, AssociativeMemory, PathIntegration) against
    ours, both fed by the reference's input functions on the reference's SSP space: identical built models, identical
    trajectories.  (Same space objects on purpose: the clean-up's argmax amplifies even a 1e-16 difference.)"""
    from sspslam_amd import harness as H
    from sspslam_amd.builder import build
    from sspslam_amd.networks import SLAMNetwork
    from oracle import OracleSimulator
    nengo, dt = ref.nengo, 0.001
    d = 55
    space_ref = ref.ssp.HexagonalSSPSpace(2, ssp_dim=d, domain_bounds=np.tile([-1.0, 1.0], (2, 1)), length_scale=0.2)
    path, vels = H.make_random_path(10.0, limit=0.2, seed=0)
    obj_locs = 0.9 * 2 * (ref.utils.Rd_sampling(5, 2, seed=0) - 0.5)
    vec_to_lm = obj_locs[None, :, :] - path[:, None, :]
    lm_space = ref.ssp.SPSpace(5, d, seed=0)
    real_ssp = space_ref.encode(path)
    vf, scale, inview, idf, spf, vecf, vecsspf = ref.slam.get_slam_input_functions2(space_ref, lm_space, vels, vec_to_lm, 0.6)

    def assemble(cls, ovc_encoders=None):                   # run_slam.py:155-195
        model = nengo.Network(seed=0)
        with model:
            vel_input = nengo.Node(vf, label="vel_input")
            init_state = nengo.Node(lambda t: real_ssp[int((t - dt) / dt)] if t < 0.05 else np.zeros(d), label="init_state")
            landmark_vec = nengo.Node(vecsspf, label="lm_vec")
            landmark_id = nengo.Node(spf, label="lm_id")
            is_landmark = nengo.Node(inview, label="lm_in_view")
            sl = cls(space_ref, lm_space, 0.6, 5, 30, 60, 20, tau_pi=0.05, update_thres=0.2, vel_scaling_factor=scale,
                     shift_rate=0.2, voja_learning_rate=1e-4, pes_learning_rate=5e-3, clean_up_method="grid", gc_n_neurons=0,
                     encoders=None, voja=True, seed=0, intercept=0.1)
            if ovc_encoders is not None:
                sl.ovc_ens.encoders = ovc_encoders
            nengo.Connection(vel_input, sl.velocity_input, synapse=None)
            nengo.Connection(init_state, sl.pathintegrator.input, synapse=None)
            nengo.Connection(landmark_vec, sl.landmark_vec_ssp, synapse=None)
            nengo.Connection(landmark_id, sl.landmark_id_input, synapse=None)
            nengo.Connection(is_landmark, sl.no_landmark_in_view, synapse=None)
            nengo.Probe(sl.pathintegrator.output, synapse=0.05)
        return model, sl

    m_ref, sl_ref = assemble(ref.slam.SLAMNetwork)
    # the reference draws the object-vector-cell encoders from the global, unseeded NumPy generator (slam.py:206);
    # ours come from RandomState(seed + 1): give both models the same ones
    m_own, sl_own = assemble(SLAMNetwork, ovc_encoders=np.asarray(sl_ref.ovc_ens.encoders))
    assert len(m_ref.all_connections) == len(m_own.all_connections) and len(m_ref.all_ensembles) == len(m_own.all_ensembles)
    # ZERO edits to the reference's class: its two Python function nodes with inputs - the clean-up lambda (slam.py:270) and
    # update_state_func (slam.py:233-237,249) - carry no `native` hint; the builder recognises them by probing
    assert sl_ref.gridcells.native is None and sl_ref.update_state.native is None
    bm_ref, bm = build(m_ref), build(m_own)
    assert [o["kind"] for o in bm_ref.ops] == [o["kind"] for o in bm.ops]
    gate_ref, gate_own = [next(o for o in m.ops if o["kind"] == "gate") for m in (bm_ref, bm)]
    assert (gate_ref["d"], gate_ref["thres"], gate_ref["rate"]) == (gate_own["d"], gate_own["thres"], gate_own["rate"]) == (d, 0.2, 0.2)
    cl_ref, cl_own = [next(o for o in m.ops if o["kind"] == "cleanup") for m in (bm_ref, bm)]
    np.testing.assert_array_equal(bm_ref.buffers[cl_ref["w"]], bm.buffers[cl_own["w"]])
    for x, y in zip(bm_ref.buffers, bm.buffers):            # (our DFT / Fourier matrices differ from the reference's in the last bits)
        np.testing.assert_allclose(np.asarray(x), np.asarray(y), atol=1e-12, rtol=0)
    a, b = OracleSimulator(bm_ref), OracleSimulator(bm)
    a.run_steps(120)
    b.run_steps(120)
    pa, pb = a.probe_data(0), b.probe_data(0)
    assert np.abs(pb).max() > 0.1
    np.testing.assert_allclose(pa, pb, atol=1e-12, rtol=0)            # observed 1.4e-16


def test_recorded_path_of_the_reference_loads():
    """``example_paths/twoRooms_path.npy`` through ``harness.load_path`` (the ``--path-data`` handling of
    run_pathint.py:77-89): cut to 49 999 rows, each axis rescaled to +-0.9, velocities by differencing; and at a
    20 ms recording step, interpolation to the 1 ms simulation step."""
    from sspslam_amd import harness as H
    f = os.path.join(REF, "example_paths", "twoRooms_path.npy")
    raw = np.load(f)
    path, vels = H.load_path(f)
    assert path.shape == (49999, 2) and vels.shape == path.shape
    np.testing.assert_allclose(path.min(0), -0.9, atol=1e-12)
    np.testing.assert_allclose(path.max(0), 0.9, atol=1e-12)
    # rescaling is affine per axis: the shape of the trajectory is the recorded one
    for i in range(2):
        c = np.corrcoef(path[:, i], raw[:49999, i])[0, 1]
        assert c > 1 - 1e-12
    np.testing.assert_allclose(np.cumsum(vels, axis=0) * 0.001 + path[0], path, atol=1e-9)
    p2, _ = H.load_path(f, data_dt=0.02, max_rows=500)
    assert p2.shape == (10000, 2)
