"""Full-size configurations and long windows on the GPU (run with -m gpu on an MI355X).

No oracle can follow BASELINE configs[3] (1.0e8 neurons) or configs[4] (a 10^6-row clean-up grid) for more than a few
timesteps, and none can follow configs[1] / [2] for seconds of simulated time.  What the domain offers instead
(size-independent properties):

* the decoded state is finite and tracks the TRUE SSP of the synthetic path (cosine similarity >= 0.99);
* graph replay and eager launches give the same bits;
* the VCO ensembles are independent (reference pathintegration.py:173-185: ensemble k's recurrence touches ensemble k
  only) - also the LAST one, whose parameters sit beyond 2^30 elements into the arrays;
* the row a clean-up returns is a maximiser of the table product for the kernel's own input;
* the f32 fast mode stays within north_star's 1e-3 cosine of the f64 parity mode (which equals the NumPy oracle to
  rounding on every window the oracle can follow) over SECONDS of simulated time - the guard against chaotic divergence
  of spiking networks.
"""
import numpy as np
import pytest

import sspslam_amd.frontend as nengo
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.fixture(scope="module")
def Simulator():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from sspslam_amd.simulator import Simulator
    return Simulator


def _similarity(out, real):
    return np.sum(out * real, axis=1) / np.maximum(np.linalg.norm(out, axis=1) * np.linalg.norm(real, axis=1), 1e-300)


def test_config4_full_size_properties(Simulator):
    """BASELINE configs[3] on one GPU: PathIntegration, ssp_dim 4033 (n_rotates 24 x n_scales 28: 2017 VCOs), 50 000 LIF
    neurons per VCO = 1.0085e8 neurons, 5.9 GB on the device; stepped by the streaming kernel (one launch per timestep)."""
    space = H.make_ssp_space(2, n_scales=28, n_rotates=24)
    assert space.ssp_dim == 4033
    K, n = 2017, 50000
    path, vels = H.make_random_path(10.0, limit=0.1, seed=0)
    pm = H.make_pathint_model(space, path, vels, n)
    with pm.model:
        p_osc = nengo.Probe(pm.pathintegrator.oscillators.output)       # (3 values per VCO: who changed is visible per VCO)
    model = build(pm.model, n_eval_points=2000)
    assert model.n_neurons == K * n
    (ens,) = [o for o in model.ops if o["kind"] == "ensarray"]
    assert (ens["K"], ens["n"]) == (K, n) and K * n * 3 > 2 ** 28          # the last VCO's encoders start > 2^30 bytes in
    steps = 232
    real = space.encode(path[:steps])
    with Simulator(None, model=model, dtype="f32") as sim:
        sim.run_steps(steps)
        out, osc = np.array(sim.data[pm.probe]), np.array(sim.data[p_osc])
        c = sim.counters()
    assert c["launches_per_step"] == 1 and c["device_bytes"] > 4 * 2 ** 30
    assert out.shape == (steps, 4033) and np.isfinite(out).all() and np.isfinite(osc).all()
    sim_true = _similarity(out, real)
    assert sim_true[60:].min() >= 0.99, sim_true[60:].min()
    # graph replay (16 timesteps per graph) == eager launches, bit for bit
    with Simulator(None, model=model, dtype="f32", steps_per_graph=1) as sim:
        sim.run_steps(32)
        np.testing.assert_array_equal(sim.data[pm.probe], out[:32])
        np.testing.assert_array_equal(sim.data[p_osc], osc[:32])
    # VCO independence: another initial voltage for the neurons of ONE oscillator changes that oscillator's outputs only -
    # the last one (highest addresses) and one in the middle
    V0 = np.zeros((K, n))
    for k in (K - 1, 1000):
        V0[:] = 0.0
        V0[k] = 0.9
        with Simulator(None, model=model, dtype="f32") as sim:
            sim.write_buffer(ens["v"], V0)
            sim.run_steps(32)
            osc2 = np.array(sim.data[p_osc])
        changed = np.any(osc2 != osc[:32], axis=0).reshape(K, 3).any(axis=1)
        assert changed[k] and changed.sum() == 1, (k, np.flatnonzero(changed)[:10])


def test_config5_full_size_properties(Simulator):
    """BASELINE configs[4] on one GPU: SLAMNetwork over a three-dimensional domain, ssp_dim=2047 -> d = 1801 (15 x 15),
    901 VCOs x 800 neurons, four 18 010-neuron populations, 20 landmarks, and the reference's 100-points-per-axis clean-up grid
    (slam.py:209): a 10^6 x 1801 table, 7.2 GB in f32 - byte offsets beyond 2^32 in one buffer."""
    space = H.make_ssp_space(3, 2047, rng=np.random.default_rng(0))
    d = space.ssp_dim
    assert d == 1801
    path, vels = H.make_random_path(10.0, limit=0.1, seed=0, domain_dim=3)
    path = path.copy()
    # start at (0.5, 0.5, .): the grid's rows are numbered axis 1 first (np.meshgrid's 'xy' order, reference sspspace.py:424-466),
    # and rows of x1 > 0.2 lie beyond 2^32 bytes in the f32 table
    path[:, 0] += 0.5 - path[0, 0]
    path[:, 1] += 0.5 - path[0, 1]
    sm = H.make_slam_model(space, path, vels, n_landmarks=20, pi_n_neurons=800, mem_n_neurons=10 * d, circonv_n_neurons=100,
                           view_rad=0.6)
    with sm.model:
        p_clean = nengo.Probe(sm.slam.gridcells)
        p_x = nengo.Probe(sm.slam.pathintegrator.output, synapse=0.01)   # the filter that feeds the clean-up (tau = 0.01)
    S = sm.slam.sample_ssps
    assert S.shape == (100 ** 3, d)
    model = build(sm.model, n_eval_points=2000)
    steps = 232
    with Simulator(None, model=model, dtype="f32") as sim:
        sim.run_steps(steps)
        out, clean, x_g = np.array(sim.data[sm.probe]), np.array(sim.data[p_clean]), np.array(sim.data[p_x])
        c = sim.counters()
    assert c["device_bytes"] > 8 * 2 ** 30 and np.isfinite(out).all() and np.isfinite(clean).all()
    sim_true = _similarity(out, sm.real_ssp[:steps])
    assert sim_true[60:].min() >= 0.99, sim_true[60:].min()
    # the pipelined step graph == one timestep's rounds launched eagerly, bit for bit
    with Simulator(None, model=model, dtype="f32", steps_per_graph=1) as sim:
        sim.run_steps(32)
        np.testing.assert_array_equal(sim.data[sm.probe], out[:32])
        np.testing.assert_array_equal(sim.data[p_clean], clean[:32])
    # the clean-up's row is a grid row and a maximiser: against the exact similarities of the kernel's OWN input (the clean-up of
    # timestep t + 1 reads the filter state left by timestep t) it is within f32 rounding of the best one
    norms = np.linalg.norm(clean, axis=1)
    assert np.abs(norms[10:] - 1.0).max() < 1e-4                         # rows of the table are unit vectors
    assert len({r.tobytes() for r in clean[100:]}) > 1                    # the cleaned-up position moves
    chosen = []
    for t in (60, 120, 180, 230):
        sims = S @ x_g[t]
        row = clean[t + 1]
        k = int(np.argmax(S @ row))                                       # the grid point the kernel returned
        np.testing.assert_allclose(row, S[k], atol=2e-6, rtol=0)
        assert (sims.max() - sims[k]) / np.linalg.norm(x_g[t]) < 2e-5, (t, k, int(np.argmax(sims)))
        chosen.append(k)
    # ... and those rows sit beyond 2^32 bytes into the f32 table (the path was moved there for this reason)
    assert min(chosen) * d * 4 > 2 ** 32, chosen


def test_long_window_config2_f32_block_kernel_tracks_f64(Simulator):
    """BASELINE configs[1], 5 simulated seconds: the f32 whole-block kernel (the headline path) against the f64 parity mode."""
    space = H.make_ssp_space(2, 1015)
    T, dt = 5.0, 0.001
    path, vels = H.make_random_path(20.0, dt=dt, limit=0.1, seed=0)
    pm = H.make_pathint_model(space, path, vels, 10000, seed=0)
    model = build(pm.model, n_eval_points=4000)
    n = int(round(T / dt))
    outs = {}
    for dtype in ("f32", "f64"):
        with Simulator(None, model=model, dtype=dtype) as sim:
            sim.run(T)                                                    # (the pipelined run a reference user makes)
            outs[dtype] = np.array(sim.data[pm.probe])
            c = sim.counters()
        assert outs[dtype].shape == (n, 1015)
        if dtype == "f32":
            assert c["launches_per_step"] == 0 and (c["block_tpb"], c["block_npt"]) == (512, 20)
    ce = H.cosine_error(outs["f32"][20:], outs["f64"][20:])
    assert ce.max() < 1e-3, ce.max()
    assert ce[:980].max() < 1e-6                                          # (first second: rounding only; observed 2e-9)
    real = space.encode(path[:n])
    assert _similarity(outs["f32"], real)[200:].mean() > 0.9              # both track the path ...
    assert abs(_similarity(outs["f32"], real)[200:].mean() - _similarity(outs["f64"], real)[200:].mean()) < 1e-3      # ... alike


def test_long_window_config3_f32_tracks_f64(Simulator):
    """BASELINE configs[2], 2 simulated seconds with the script's landmark positions (landmarks enter and leave the view, PES and
    Voja learn): f32 fast mode against the f64 parity mode - trajectory, learned decoders through map recall (i)."""
    import sspslam_amd.frontend as fe
    sm = H.make_config3_model(landmark_near_start=False)
    model = build(sm.model, n_eval_points=4000)
    am = sm.slam.assomemory
    wb = model.params[am.conn_out].learned_buffer
    steps = 2000
    outs, W = {}, {}
    for dtype in ("f32", "f64"):
        with Simulator(None, model=model, dtype=dtype) as sim:
            sim.run_steps(steps)
            outs[dtype] = np.array(sim.data[sm.probe])
            W[dtype] = sim.read_buffer(wb)
    ce = H.cosine_error(outs["f32"][20:], outs["f64"][20:])
    assert ce.max() < 1e-3, ce.max()
    assert np.abs(W["f64"]).max() > 0                                     # a landmark was seen: PES learned something
    rec32, _ = H.map_recall(sm.ssp_space, sm.lm_space, model.params[am.memory], fe.LIF(), W["f32"])
    rec64, _ = H.map_recall(sm.ssp_space, sm.lm_space, model.params[am.memory], fe.LIF(), W["f64"])
    nr = np.linalg.norm(rec64, axis=1)
    seen = nr > 0.05 * nr.max()            # (a landmark glimpsed for a few timesteps recalls a vector of norm ~1e-5: rounding noise)
    assert seen.any() and H.cosine_error(rec32[seen], rec64[seen]).max() < 1e-3
    assert _similarity(outs["f32"], sm.real_ssp[:steps])[200:].mean() > 0.9
