"""The oracle's nengo-semantics restatement has no fixture in the reference (it has no tests and nengo is
absent): these tests pin it to analytic properties of the published algorithm instead."""
import numpy as np
import pytest

import sspslam_amd.frontend as nengo
from sspslam_amd.builder import build
from oracle import OracleSimulator, lif_rate, lif_step

from helpers import small_pathint


def test_lif_step_matches_rate_curve():
    """Spike count of the spiking LIF under constant current converges to the LIFRate formula (A.4)."""
    dt, tau_rc, tau_ref = 0.001, 0.02, 0.002
    J = np.array([0.5, 1.0, 1.2, 2.0, 5.0, 20.0])
    V, R = np.zeros_like(J), np.zeros_like(J)
    count = np.zeros_like(J)
    steps = 20000
    for _ in range(steps):
        count += lif_step(J, V, R, dt, tau_rc, tau_ref, 0.0)
    want = lif_rate(J.copy(), tau_rc, tau_ref)
    np.testing.assert_allclose(count / (steps * dt), want, rtol=2e-3, atol=0.06)
    assert count[0] == 0 and count[1] == 0          # J <= 1 never spikes


def test_lif_step_single_step_by_hand():
    dt, tau_rc, tau_ref = 0.001, 0.02, 0.002
    J = np.array([30.0, 0.5, 3.0])
    V = np.array([0.9, 0.2, 0.5])
    R = np.array([0.0, 0.0, 0.0015])      # third neuron is refractory for another 0.5 ms after this step starts
    spiked = lif_step(J, V, R, dt, tau_rc, tau_ref, 0.0)
    # neuron 0: full step from 0.9 towards 30 crosses threshold
    v0 = 0.9 - (30 - 0.9) * np.expm1(-dt / tau_rc)
    t_spike = dt + tau_rc * np.log1p(-(v0 - 1) / (30 - 1))
    assert spiked.tolist() == [True, False, False]
    assert V[0] == 0 and np.isclose(R[0], tau_ref + t_spike) and 0 < t_spike < dt
    assert np.isclose(V[1], 0.2 - (0.5 - 0.2) * np.expm1(-dt / tau_rc))
    # neuron 2 integrates only for the 0.5 ms left after its refractory period ends
    assert np.isclose(R[2], 0.0005) and np.isclose(V[2], 0.5 - (3 - 0.5) * np.expm1(-0.0005 / tau_rc))


def test_lowpass_and_one_step_delay():
    """Node -> Lowpass(tau) -> Node -> probe: y_t = a*y_{t-1} + (1-a)*u_t, visible one step later (A.5/A.6)."""
    dt, tau = 0.001, 0.01
    with nengo.Network(seed=0) as m:
        u = nengo.Node(lambda t: 1.0)
        a = nengo.Node(size_in=1)
        b = nengo.Node(size_in=1)
        nengo.Connection(u, a, synapse=None)          # same step
        nengo.Connection(a, b, synapse=tau)           # filtered, seen next step
        pa = nengo.Probe(a)
        pb = nengo.Probe(b)
        pf = nengo.Probe(a, synapse=tau)              # probe filter: includes this step's input
    sim = OracleSimulator(build(m, dt=dt))
    sim.run_steps(50)
    ya, yb, yf = sim.probe_data(0)[:, 0], sim.probe_data(1)[:, 0], sim.probe_data(2)[:, 0]
    al = np.exp(-dt / tau)
    filt = 1 - al ** np.arange(1, 51)
    np.testing.assert_allclose(ya, 1.0)
    np.testing.assert_allclose(yf, filt, atol=1e-14)
    np.testing.assert_allclose(yb[1:], filt[:-1], atol=1e-14)
    assert yb[0] == 0.0
    np.testing.assert_allclose(sim.trange(), dt * np.arange(1, 51))


def test_node_time_and_transform():
    with nengo.Network(seed=0) as m:
        u = nengo.Node(lambda t: [t, 2 * t])
        v = nengo.Node(size_in=3)
        nengo.Connection(u, v, transform=np.array([[1.0, 0], [0, 1.0], [1.0, 1.0]]), synapse=None)
        nengo.Connection(u[1], v[0], transform=-0.5, synapse=None)
        p = nengo.Probe(v)
    sim = OracleSimulator(build(m))
    sim.run_steps(3)
    t = 0.001 * np.arange(1, 4)
    np.testing.assert_allclose(sim.probe_data(0), np.stack([t - t, 2 * t, 3 * t], 1), atol=1e-15)


def test_ensemble_represents_and_decodes():
    """A 1-D LIF ensemble decodes its (filtered) input within NEF accuracy; rate and ReLU types run too."""
    for nt, tol in ((nengo.LIF(), 0.06), (nengo.LIFRate(), 0.03), (nengo.RectifiedLinear(), 0.03)):
        with nengo.Network(seed=2) as m:
            u = nengo.Node(lambda t: 0.6)
            e = nengo.Ensemble(200, 1, neuron_type=nt)
            nengo.Connection(u, e, synapse=None)
            p = nengo.Probe(e, synapse=0.03)
        sim = OracleSimulator(build(m))
        sim.run_steps(400)
        assert abs(sim.probe_data(0)[-100:, 0].mean() - 0.6) < tol, type(nt).__name__


def test_vco_oscillates_at_commanded_frequency():
    """One VCO ensemble driven with omega = A.v/l keeps unit radius and rotates at that rate."""
    pm = small_pathint(ssp_dim=55, n=300, T=10.0, limit=0.2, seed=3)
    model = build(pm.model)
    sim = OracleSimulator(model)
    sim.run_steps(1200)
    out = sim.probe_data(0)
    from sspslam_amd import harness as H
    est, sims, err = H.pathint_metrics(pm.ssp_space, out, pm.real_ssp, pm.path)
    # the probe is Lowpass(50 ms)-filtered, so it lags the instantaneous truth a little
    assert sims[200:].mean() > 0.85 and sims[200:].min() > 0.7        # tracks the true SSP
    assert np.median(err[200:]) < 0.08


def test_pes_learns_and_voja_moves_encoders():
    from sspslam_amd.networks import AssociativeMemory
    d = 8
    rng = np.random.RandomState(0)
    key = rng.randn(d)
    key /= np.linalg.norm(key)
    val = rng.randn(d)
    val *= 0.5 / np.linalg.norm(val)
    with nengo.Network(seed=1) as m:
        k = nengo.Node(lambda t: key)
        v = nengo.Node(lambda t: val)
        gate = nengo.Node(lambda t: 0.0)
        am = AssociativeMemory(300, d, d, intercept=0.2, voja_learning_rate=5e-3, pes_learning_rate=1e-3)
        nengo.Connection(k, am.key_input, synapse=None)
        nengo.Connection(v, am.value_input, synapse=None)
        nengo.Connection(gate, am.learning, synapse=None)
        p = nengo.Probe(am.recall, synapse=0.05)
        pw = nengo.Probe(am.conn_out, "weights", sample_every=0.1)
    model = build(m)
    sim = OracleSimulator(model)
    enc0 = model.buffers[model.params[am.memory].encoder_buffer].copy()
    sim.run_steps(3000)
    rec = sim.probe_data(0)
    cos = rec[-1] @ val / np.linalg.norm(rec[-1]) / np.linalg.norm(val)
    assert cos > 0.9                                            # recall converges to the value
    enc1 = sim.buf[model.params[am.memory].encoder_buffer]
    moved = np.linalg.norm(enc1 - enc0, axis=1) > 0
    assert 0 < moved.sum() < 300                                # only neurons that spiked moved
    sim0 = (enc0[moved] / np.linalg.norm(enc0[moved], axis=1, keepdims=True)) @ key
    sim1 = (enc1[moved] / np.linalg.norm(enc1[moved], axis=1, keepdims=True)) @ key
    assert sim1.mean() > sim0.mean()                            # ... towards the key
    W = sim.probe_data(1)
    assert W.shape == (30, d, 300) and np.abs(W[-1]).max() > 0


def test_gate_and_cleanup_ops():
    from sspslam_amd import HexagonalSSPSpace
    from sspslam_amd.networks.slam import make_gate
    s = HexagonalSSPSpace(2, ssp_dim=7, domain_bounds=np.tile([-1.0, 1.0], (2, 1)), length_scale=0.3)
    grid, _ = s.get_sample_pts_and_ssps(10)
    x = s.encode(np.array([[0.2, -0.3]]))[0]
    with nengo.Network(seed=0) as m:
        u = nengo.Node(lambda t: x + 0.05 * np.sin(40 * t))
        c = nengo.Node(lambda t, v: grid[np.argmax(grid @ v)], size_in=7, size_out=7)
        c.native = ("cleanup", grid)
        nengo.Connection(u, c, synapse=None)
        g = nengo.Node(make_gate(7, 0.2, 0.5), size_in=15, size_out=7)
        g.native = ("gate", 7, 0.2, 0.5)
        nengo.Connection(c, g[:7], synapse=None)
        nengo.Connection(u, g[7:14], synapse=None)
        flag = nengo.Node(lambda t: 0.0 if t < 0.01 else 10.0)
        nengo.Connection(flag, g[14], synapse=None)
        pc, pg = nengo.Probe(c), nengo.Probe(g)
    sim = OracleSimulator(build(m))
    sim.run_steps(20)
    C, G = sim.probe_data(0), sim.probe_data(1)
    for i in range(20):
        t = 0.001 * (i + 1)
        xin = x + 0.05 * np.sin(40 * t)
        want_c = grid[np.argmax(grid @ xin)]
        np.testing.assert_allclose(C[i], want_c, atol=1e-15)
        want_g = 0.5 * (want_c - xin) if (t < 0.01 and want_c @ xin > 0.2) else np.zeros(7)
        np.testing.assert_allclose(G[i], want_g, atol=1e-15)
    assert np.abs(G[:9]).max() > 0 and np.abs(G[10:]).max() == 0


def test_slamview_model_runs_on_the_oracle_and_io_helpers(tmp_path):
    """SLAMViewNetwork (reference slam_view.py / run_slamview.py) lowers to the same operator set minus the
    circular convolutions and steps on the oracle; recorded-trajectory loading and the result files use the
    reference scripts' formats (run_pathint.py:57-89,201-207; run_slam.py:282-293)."""
    from sspslam_amd import harness as H
    from sspslam_amd.builder import build
    s = H.make_ssp_space(2, 55)
    d = s.ssp_dim
    path, vels = H.make_random_path(10.0, limit=0.2, seed=0)
    sm = H.make_slamview_model(s, path, vels, n_landmarks=5, pi_n_neurons=30, mem_n_neurons=60, view_rad=0.6,
                               weights_sample_every=0.05)
    model = build(sm.model)
    kinds = {o["kind"] for o in model.ops}
    assert {"cleanup", "gate", "pes", "voja", "neurons", "ensarray"} <= kinds
    ref = OracleSimulator(model)
    ref.run_steps(200)
    out = ref.probe_data(0)
    assert out.shape == (200, d) and np.isfinite(out).all() and np.abs(out).max() > 0.1
    assert np.abs(ref.probe_data(2)[-1]).max() > 0          # PES moved the decoders

    # recorded trajectory: 50 Hz -> 1 kHz, rescaled to +-0.9
    rec = np.cumsum(np.random.RandomState(0).randn(300, 2), axis=0)
    np.save(tmp_path / "rec.npy", rec)
    p2, v2 = H.load_path(str(tmp_path / "rec.npy"), data_dt=0.02)
    assert p2.shape == (6000, 2) and abs(p2.max() - 0.9) < 1e-12 and abs(p2.min() + 0.9) < 1e-12
    np.testing.assert_allclose(v2[1:], np.diff(p2, axis=0) / 0.001)
    st = H.stretch_trajectory(rec, 0.02, 0.001)
    np.testing.assert_allclose(st[0], rec[0]); np.testing.assert_allclose(st[-1], rec[-1])

    real = s.encode(path)
    est = H.save_pathint_results(str(tmp_path / "pi.npz"), s, np.arange(1, 201) * 0.001, path, real, out, 1.5)
    z = np.load(tmp_path / "pi.npz", allow_pickle=True)
    assert {"ts", "path", "real_ssp", "pi_sim_out", "pi_sims", "pi_path", "pi_error", "elapsed_time",
            "elapsed_thread_time", "args", "sig_to_noise_ratio"} <= set(z.files)
    assert z["pi_path"].shape == (200, 2) and est.shape == (200, 2) and z["pi_error"].shape == (200,)
    H.save_slam_results(str(tmp_path / "slam.npz"), s, np.arange(1, 201) * 0.001, path, real, sm.obj_locs, 0.6, out,
                        np.zeros((5, d)), np.zeros((5, 2)), 1.5)
    z = np.load(tmp_path / "slam.npz", allow_pickle=True)
    assert {"timesteps", "slam_sim_out", "slam_sims", "slam_path", "slam_error", "landmark_ssps_est",
            "landmark_loc_est", "obj_locs", "view_rad"} <= set(z.files)


def test_vectorised_input_tables_equal_the_per_step_closures():
    """harness.indexed_rows_node_fn: the `.table(steps)` twin used by the HIP simulator's tabulation returns, step for
    step, the rows the reference's closure `lambda t: table[int((t - dt) / dt)]` returns (float64 time arithmetic
    included: int((t - dt) / dt) is not always step - 1)."""
    from sspslam_amd import harness as H
    from sspslam_amd.simulator import tabulate
    dt = 0.001
    table = np.random.RandomState(0).randn(30001, 3)
    for until in (None, 0.05):
        fn = H.indexed_rows_node_fn(table, dt, until=until)
        steps = np.arange(1, 30001)
        rows, idx = fn.table(steps)
        got = np.where(idx[:, None] >= 0, rows[np.maximum(idx, 0)], 0.0) if len(rows) else np.zeros((len(steps), 3))
        plain = lambda t: fn(t)                                      # no .table attribute: the per-step path
        rows2, idx2 = tabulate(plain, 3, steps, dt)
        got2 = np.where(idx2[:, None] >= 0, rows2[np.maximum(idx2, 0)], 0.0) if len(rows2) else np.zeros((len(steps), 3))
        np.testing.assert_array_equal(got, got2)
        if until is not None:
            assert (idx2[steps * dt >= until] == -1).all() and len(rows2) < 60      # zero rows are not stored
        k = np.array([int((t - dt) / dt) for t in (steps * dt).tolist()])
        assert until is not None or (k != steps - 1).any()           # the quirk is exercised
        # a later chunk only (what the pipelined tabulation asks for)
        r3, i3 = fn.table(np.arange(2049, 4097))
        np.testing.assert_array_equal(np.where(i3[:, None] >= 0, r3[np.maximum(i3, 0)], 0.0) if len(r3) else np.zeros((2048, 3)),
                                      got[2048:4096])


def test_plain_closure_tabulation_is_exact_and_copies_at_call_time():
    """simulator.tabulate on closures without a `.table` twin: one call per timestep in time order, the value copied before
    the next call (a closure may hand out ONE buffer again and again), consecutive equal rows stored once, zero rows not at
    all - and the rows it reports are, bit for bit, what per-step calls return."""
    from sspslam_amd.simulator import tabulate, RowStage
    dt = 0.001
    rng = np.random.RandomState(3)
    table = rng.randn(5000, 7)
    table[100:140] = table[100]                      # a stretch of equal rows
    table[200:260] = 0.0                             # zeros inside the table
    table[300, 3] = -0.0
    table[301] = table[300]
    table[301, 3] = 0.0                              # differs from its neighbour in the sign of a zero only
    calls = []
    buf = np.empty(7)

    def reusing(t):                                  # hands out the same array object every time
        calls.append(t)
        buf[:] = table[int((t - dt) / dt)] if t < 4.0 else 0.0
        return buf

    steps = np.arange(1, 4501)
    stage = RowStage()
    rows, idx = tabulate(reusing, 7, steps, dt, stage)
    assert calls[2:] == (steps * dt).tolist() and calls[:2] == [steps[0] * dt, steps[-1] * dt]      # two size checks, then time order
    want = np.stack([table[int((t - dt) / dt)] if t < 4.0 else np.zeros(7) for t in (steps * dt).tolist()])
    got = np.where(idx[:, None] >= 0, rows[np.maximum(idx, 0)], 0.0)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert (idx[3999:] == -1).all() and (idx[200:260] == -1).all()
    assert idx[300] != idx[301]                      # -0.0 and 0.0 are different rows
    assert len(set(idx[100:140].tolist())) == 1
    # a narrow, mostly constant node goes the compacting way; a scalar-returning function for a wider node is refused
    rows2, idx2 = tabulate(lambda t: [1.0, 2.0] if t < 2.0 else [3.0, 4.0], 2, steps, dt)
    assert rows2.tolist() == [[1.0, 2.0], [3.0, 4.0]] and idx2[:1999].tolist() == [0] * 1999 and (idx2[1999:] == 1).all()
    with pytest.raises(Exception):
        tabulate(lambda t: 1.0, 3, steps, dt)


def test_chunk_schedule_of_a_pipelined_run():
    """Simulator._next_chunk_len: chunks cover the run exactly, grow only as fast as the measured tabulation hides behind the
    device, stay inside [PIPELINE_MIN, PIPELINE_MAX] and leave a short last chunk (its read-back is not overlapped)."""
    from sspslam_amd.simulator import Simulator
    for dev, tab in ((3e-6, 0.1e-6), (3e-6, 2.0e-6), (3e-6, 6e-6), (None, None)):
        sim = Simulator.__new__(Simulator)
        sim._dev_rate, sim._tab_rate = dev, tab
        for steps in (5000, 20000, 20001, 100000):
            chunks = [min(Simulator.PIPELINE_FIRST, steps)]
            left = steps - chunks[0]
            while left > 0:
                n = sim._next_chunk_len(chunks[-1], left)
                assert 1 <= n <= left
                chunks.append(n)
                left -= n
            assert sum(chunks) == steps
            assert max(chunks) <= Simulator.PIPELINE_MAX
            assert chunks[-1] <= Simulator.PIPELINE_TAIL and (len(chunks) < 3 or chunks[-2] <= Simulator.PIPELINE_MID + Simulator.PIPELINE_TAIL)
            if dev and tab and tab > dev:            # tabulation-bound: chunks shrink to the minimum instead of stalling the device longer
                assert sorted(chunks[1:-2])[len(chunks[1:-2]) // 2] <= Simulator.PIPELINE_FIRST
            if dev and tab and tab < 0.2 * dev:
                assert len(chunks) <= 5 + steps // Simulator.PIPELINE_MAX
