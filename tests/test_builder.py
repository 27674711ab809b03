"""Host logic: lowering, operator merging, scheduling rule, pruning and VCO sharding of the builder."""
import numpy as np
import pytest

import sspslam_amd.frontend as nengo
from sspslam_amd.builder import build, op_access, _overlap
from sspslam_amd.networks import CircularConvolution, circconv
from oracle import OracleSimulator

from helpers import small_pathint


@pytest.fixture(scope="module")
def pi_model():
    pm = small_pathint(ssp_dim=55, n=40, T=10.0, limit=0.2)
    return pm, build(pm.model)


def test_pi_lowers_to_a_handful_of_merged_ops(pi_model):
    pm, m = pi_model
    kinds = [o["kind"] for o in m.ops]
    assert kinds.count("ensarray") == 1 and kinds.count("matvec") == 3       # to_Fourier, velocity (stacked), to_SSP
    assert len(m.ops) <= 40 and m.stats["n_raw_ops"] > 100                    # 27 VCOs' worth of ops merged
    ens = next(o for o in m.ops if o["kind"] == "ensarray")
    assert (ens["K"], ens["n"], ens["din"], ens["dout"]) == (28, 40, 3, 4)
    vel = [o for o in m.ops if o["kind"] == "matvec" and o["cols"] == 2]
    assert len(vel) == 1 and vel[0]["rows"] == 81                            # 27 stacked 3x2 velocity transforms
    # the third (omega) row of the feedback function is identically zero and was dropped; so was the omega row of the
    # identity output: to_SSP (reference get_from_Fourier, pathintegration.py:824-844) has a zero column for it and
    # nothing else reads or probes it (dead decoded rows, builder._live_elements)
    dec = m.buffers[ens["dec"]]
    assert dec.shape == (28, 4, 40) and np.all(dec[0, 1:] == 0) and np.any(dec[1, 3] != 0)
    assert m.stats["dead_decoded_rows"] == 29            # 28 frequency rows + the imaginary part of the zero-frequency VCO


def test_a_probed_or_used_decoded_row_stays(pi_model):
    """The frequency dimension of oscillators.output is dropped only while nothing can observe it."""
    pm = small_pathint(ssp_dim=55, n=40, T=10.0, limit=0.2)
    with pm.model:
        nengo.Probe(pm.pathintegrator.oscillators.output[3:6], synapse=None)
    m = build(pm.model)
    ens = next(o for o in m.ops if o["kind"] == "ensarray")
    assert ens["dout"] == 5 and m.stats["dead_decoded_rows"] == 28


def test_schedule_respects_set_inc_read_update(pi_model):
    _, m = pi_model
    acc = [op_access(o, m) for o in m.ops]
    for i in range(len(m.ops)):
        for j in range(i + 1, len(m.ops)):
            for ci in range(4):
                for cj in range(4):
                    if ci == 2 and cj == 2:
                        continue
                    if any(_overlap(a, b) for a in acc[i][ci] for b in acc[j][cj]):
                        assert ci <= cj, (m.ops[i]["kind"], ci, m.ops[j]["kind"], cj)
    # micro ops of one level touch disjoint data
    for i in range(len(m.ops)):
        for j in range(i + 1, len(m.ops)):
            if m.ops[i]["level"] == m.ops[j]["level"]:
                wi = acc[i][0] + acc[i][1] + acc[i][3]
                wj = acc[j][0] + acc[j][1] + acc[j][3]
                assert not any(_overlap(a, b) for a in wi for b in wj + acc[j][2])
                assert not any(_overlap(a, b) for a in wj for b in acc[i][2])


def test_stage_partition_of_pathintegration(pi_model):
    """Only the VCO array and its feedback filter are stepped per timestep; the to_Fourier input chain
    (pre) and the to_SSP read-out chain (post) are feed-forward in time and run time-batched."""
    _, m = pi_model
    info = m.stage_info
    assert info["enabled"] and info["n_core"] == 3
    core = [o["kind"] for o in m.ops if o["stage"] == 1]
    assert sorted(core) == ["axpy", "ensarray", "lowpass"]
    pre = [o for o in m.ops if o["stage"] == 0]
    post = [o for o in m.ops if o["stage"] == 2]
    assert sum(o["kind"] == "matvec" for o in pre) == 2 and sum(o["kind"] == "matvec" for o in post) == 1
    assert [o["stage"] for o in m.ops] == sorted(o["stage"] for o in m.ops)          # stage-major order
    ens = next(o for o in m.ops if o["kind"] == "ensarray")
    assert info["pre_to_core"] == [(ens["x"], ens["x"] + 84)]                           # the VCO inputs
    # (one range although the dead frequency rows leave holes in what the core writes; the last VCO's hole is cut off)
    assert len(info["core_to_post"]) == 1 and info["core_to_post"][0][1] - info["core_to_post"][0][0] == 83
    assert info["probe_stage"] == [2]
    # batched order is a permutation; pre-update state reads are flagged
    for st in (pre, post):
        assert sorted(o["border"] for o in st) == list(range(len(st)))
    flagged = [o for o in m.ops if o["src_prev"]]
    assert len(flagged) >= 2 and all(o["kind"] == "axpy" for o in flagged)
    # staged and unstaged orders give the same oracle trajectory
    pm2 = small_pathint(ssp_dim=55, n=40, T=10.0, limit=0.2)
    a, b = OracleSimulator(m), OracleSimulator(build(pm2.model, staged=False))
    a.run_steps(120)
    b.run_steps(120)
    np.testing.assert_allclose(a.probe_data(0), b.probe_data(0), atol=1e-13)


def _stage_order_holds(m):
    """Every writer of a signal runs in a stage no later than every reader of it (any lag): the core stage steps per timestep,
    the post stage once per block behind it - a core reader of a state that the post stage updates would see a stale value."""
    acc = [op_access(o, m) for o in m.ops]
    for i, a in enumerate(acc):
        w = [r for cls in (0, 1, 3) for r in a[cls]]
        for j, b in enumerate(acc):
            if i != j and any(_overlap(x, y) for x in w for y in b[2]):
                assert m.ops[i]["stage"] <= m.ops[j]["stage"], (m.ops[i], m.ops[j])


def test_stage_order_when_one_hand_off_reads_core_and_read_out_filters(pi_model):
    """A dense population with two decoded connections, one of them only ever read by a probe: the builder merges the copies
    of both filters' states into the node inputs into one operator; the synapse rule moves that operator into the core stage
    (it reads a core filter's state before the update), and the read-out filter it also reads - with the decode that feeds it -
    has to follow (round 4: left in the post stage, the second connection read zeros on the device)."""
    d = 19
    with nengo.Network(seed=4) as net:
        u = nengo.Node(lambda t: np.sin(8 * t + np.arange(d)) * 0.8)
        a = nengo.Ensemble(200, d)
        e = nengo.Ensemble(300, d)
        o = nengo.Node(size_in=2)
        o2 = nengo.Node(size_in=1)
        nengo.Connection(u, a, synapse=None)
        nengo.Connection(a, e, synapse=0.01)
        nengo.Connection(e, o, synapse=0.01, function=lambda x: [x[0] * x[1], x[0]])
        nengo.Connection(e, o2, synapse=0.005, function=lambda x: x[0] ** 2)
        nengo.Probe(o, synapse=0.02)
        nengo.Probe(o2)
    m = build(net)
    assert m.stage_info["enabled"]
    _stage_order_holds(m)
    readout = [o_ for o_ in m.ops if o_["kind"] == "matvec" and o_["cols"] == 300]
    assert len(readout) == 2 and all(o_["stage"] == 1 for o_ in readout)
    _stage_order_holds(pi_model[1])


def test_cycle_without_synapse_is_rejected():
    with nengo.Network(seed=0) as m:
        a = nengo.Node(size_in=1)
        b = nengo.Node(size_in=1)
        nengo.Connection(a, b, synapse=None)
        nengo.Connection(b, a, synapse=None)
    with pytest.raises(nengo.BuildError):
        build(m)


def test_unsupported_python_node_is_rejected():
    with nengo.Network(seed=0) as m:
        u = nengo.Node(lambda t: [1.0, 2.0])
        f = nengo.Node(lambda t, x: x ** 2, size_in=2, size_out=2)
        nengo.Connection(u, f, synapse=None)
    with pytest.raises(nengo.BuildError, match="native"):
        build(m)


def test_identity_function_node_is_recognised():
    with nengo.Network(seed=0) as m:
        u = nengo.Node(lambda t: [t, -t])
        f = nengo.Node(lambda t, x: x, size_in=2, size_out=2)
        nengo.Connection(u, f, synapse=None)
        p = nengo.Probe(f)
    sim = OracleSimulator(build(m))
    sim.run_steps(2)
    np.testing.assert_allclose(sim.probe_data(0), [[0.001, -0.001], [0.002, -0.002]])


def test_seeded_build_is_reproducible():
    a = build(small_pathint(ssp_dim=7, n=20).model)
    b = build(small_pathint(ssp_dim=7, n=20).model)
    for x, y in zip(a.buffers, b.buffers):
        np.testing.assert_array_equal(x, y)


def test_vco_shards_union_is_the_model():
    """Sharded builds own disjoint VCO ranges with identical parameters; pruning drops the read-out."""
    full = build(small_pathint(ssp_dim=55, n=30, T=10.0, limit=0.2).model)
    ens_f = next(o for o in full.ops if o["kind"] == "ensarray")
    got = []
    for r in range(3):
        pm = small_pathint(ssp_dim=55, n=30, T=10.0, limit=0.2)
        m = build(pm.model, vco_shard=(r, 3))
        e = next(o for o in m.ops if o["kind"] == "ensarray")
        assert e["k_total"] == 28 and e["K"] == (10 if r < 2 else 8) and e["k_lo"] == 10 * r
        np.testing.assert_array_equal(m.buffers[e["enc"]], full.buffers[ens_f["enc"]][e["k_lo"]:e["k_lo"] + e["K"]])
        np.testing.assert_array_equal(m.buffers[e["dec"]], full.buffers[ens_f["dec"]][e["k_lo"]:e["k_lo"] + e["K"]])
        got.append(e["K"])
        assert m.sig_size == full.sig_size               # identical signal layout on every rank
    assert sum(got) == 28


def test_circconv_network_computes_binding():
    """The neural circular convolution approximates circconv(a, b) (the in-tree NumPy oracle, binding.py:12)."""
    d = 8
    rng = np.random.RandomState(1)
    a, b = rng.randn(2, d)
    a /= np.linalg.norm(a)
    b /= np.linalg.norm(b)
    with nengo.Network(seed=3) as m:
        na, nb = nengo.Node(lambda t: a), nengo.Node(lambda t: b)
        cc = CircularConvolution(200, d)
        nengo.Connection(na, cc.input_a, synapse=None)
        nengo.Connection(nb, cc.input_b, synapse=None)
        p = nengo.Probe(cc.output, synapse=0.03)
    model = build(m)
    assert sum(o["kind"] == "ensarray" for o in model.ops) == 2      # sq1 and sq2 arrays
    sim = OracleSimulator(model)
    sim.run_steps(300)
    got = sim.probe_data(0)[-100:].mean(0)
    want = circconv(a, b)
    assert got @ want / np.linalg.norm(got) / np.linalg.norm(want) > 0.97


def test_cleanup_operator_carries_the_grid_factors():
    """A clean-up node tagged by our networks hands the device the factor tables of its sample grid (three extra
    buffers on the operator, shapes (2K, d), (n, 2K), (n^(dim-1), 2K)); a node tagged without factors, or whose table is
    not that grid, gets the table only."""
    from sspslam_amd import harness as H
    s = H.make_ssp_space(2, 55)
    table, _ = s.get_sample_pts_and_ssps(100)
    gf = s.grid_factors(100)

    def net(native):
        with nengo.Network(seed=0) as m:
            u = nengo.Node(lambda t: table[17] * np.cos(t))
            c = nengo.Node(lambda t, x: table[np.argmax(table @ x)], size_in=55, size_out=55)
            if native is not None:
                c.native = native
            nengo.Connection(u, c, synapse=0.01)
            nengo.Probe(c)
        return build(m)

    model = net(("cleanup", table, gf))
    op = next(o for o in model.ops if o["kind"] == "cleanup")
    assert (op["grid_rows"], op["grid_cols"], op["grid_k2"]) == (100, 100, 56)
    assert [model.buffers[op[k]].shape for k in ("g_dft", "g_lhs", "g_rhs")] == [(56, 55), (100, 56), (100, 56)]
    assert len(op_access(op, model)[2]) == 5                      # reads: source range, table and the three factor buffers
    for native in (("cleanup", table), ("cleanup", table[:5000], gf)):
        op = next(o for o in net(native).ops if o["kind"] == "cleanup")
        assert "g_dft" not in op and op["cols"] == 55



def test_glue_collapse_keeps_the_trajectory_and_cuts_the_levels():
    """builder.collapse (glue.py): chains of fill / axpy / copy operators between the big operators of the per-timestep
    core are folded into lincomb operators.  Same trajectory, decoders and encoders as the uncollapsed list (the oracle
    executes both; sums are re-associated, so equal to rounding), fewer dependency levels, and no axpy left that reads
    an accumulator only glue wrote."""
    from sspslam_amd import harness as H
    from oracle import OracleSimulator
    s = H.make_ssp_space(2, 55)
    path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
    res = {}
    for collapse in (False, True):
        sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons=120, circonv_n_neurons=30, view_rad=0.6)
        m = build(sm.model, collapse=collapse)
        core = [o for o in m.ops if o["stage"] == 1]
        sim = OracleSimulator(m)
        sim.run_steps(150)
        am = sm.slam.assomemory
        res[collapse] = (sim.probe_data(0), sim.buf[m.params[am.conn_out].learned_buffer].copy(),
                         sim.buf[m.params[am.memory].encoder_buffer].copy(), len({o["level"] for o in core}), core, m.stats)
    np.testing.assert_allclose(res[True][0], res[False][0], atol=1e-12, rtol=0)
    np.testing.assert_allclose(res[True][1], res[False][1], atol=1e-14, rtol=1e-10)
    np.testing.assert_allclose(res[True][2], res[False][2], atol=1e-13, rtol=1e-10)
    assert np.abs(res[True][1]).max() > 0
    assert res[True][3] <= res[False][3] - 5, (res[True][3], res[False][3])
    assert any(o["kind"] == "lincomb" for o in res[True][4]) and not any(o["kind"] == "lincomb" for o in res[False][4])
    st = res[True][5]
    assert st["glue_inlined_segments"] > 0 and st["glue_glue_out"] < st["glue_glue_in"]


def test_two_pass_through_nodes_joined_by_synapses_in_both_directions():
    """ADVICE r2 (glue.py copy propagation): a -> b through Lowpass(10 ms) and b -> a through Lowpass(20 ms), a and b
    pass-through nodes.  Retargeting both filters onto each other's STATE leaves the scheduler a cycle (each must read the
    other's state before that state is updated); the copy has to stay.  Both builds step to the same trajectory."""
    def net():
        with nengo.Network(seed=1) as m:
            drive = nengo.Node(lambda t: [np.sin(8 * t), np.cos(5 * t)])
            a = nengo.Node(size_in=2, label="a")
            b = nengo.Node(size_in=2, label="b")
            nengo.Connection(drive, a, synapse=None)
            nengo.Connection(a, b, synapse=0.01)
            nengo.Connection(b, a, synapse=0.02, transform=0.5)
            pa, pb = nengo.Probe(a), nengo.Probe(b)
        return m
    res = {}
    for collapse in (False, True):
        model = build(net(), collapse=collapse)             # (used to raise "operator graph has a cycle within one timestep")
        sim = OracleSimulator(model)
        sim.run_steps(200)
        res[collapse] = (sim.probe_data(0), sim.probe_data(1))
    assert np.abs(res[False][1]).max() > 0.05
    for x, y in zip(res[True], res[False]):
        np.testing.assert_allclose(x, y, atol=1e-13, rtol=0)
    # the retargeting itself still happens where no such cycle exists (the SLAM network: 10 inputs)
    from sspslam_amd import harness as H
    s = H.make_ssp_space(2, 55)
    path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
    sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons=120, circonv_n_neurons=30, view_rad=0.6)
    assert build(sm.model).stats["glue_retargeted_inputs"] >= 10


def test_neuron_shard_refuses_what_it_cannot_split():
    """ADVICE r2 (builder.shard_phases): a partial sum must never reach a non-linearity un-exchanged, and shapes the shard
    builder cannot slice are refused with a BuildError instead of an IndexError deep in the lowering."""
    def net(kind):
        with nengo.Network(seed=3) as m:
            u = nengo.Node(lambda t: [np.sin(6 * t), np.cos(6 * t)])
            a = nengo.Ensemble(40, 2, label="a")
            b = nengo.Ensemble(30, 2, label="b")
            nengo.Connection(u, a, synapse=None)
            if kind == "decoded into neurons":           # Ensemble -> other.neurons through a 30 x 2 matrix
                nengo.Connection(a, b.neurons, transform=np.ones((30, 2)) * 0.1, synapse=0.01)
            elif kind == "unfiltered decode into neurons":   # a's partial decoded sum reaches b's currents within the timestep
                nengo.Connection(a, b, synapse=None)
            nengo.Probe(b, synapse=0.01)
        return m
    with pytest.raises(nengo.BuildError, match="neurons of a neuron-sharded ensemble"):
        build(net("decoded into neurons"), neuron_shard=(0, 2))
    with pytest.raises(nengo.BuildError, match="partial sum"):
        build(net("unfiltered decode into neurons"), neuron_shard=(0, 2))
    # ... and both are fine when the reading ensemble's source is replicated or the model is not sharded
    build(net("decoded into neurons"))
    m = net("unfiltered decode into neurons")
    build(m, neuron_shard=(0, 2), replicate=[m.ensembles[0]])


def test_function_nodes_are_recognised_without_hints():
    """builder._probe_function_node: the reference's clean-up lambda (slam.py:212-215,270) and gate (slam.py:233-237,249)
    are recognised from closures + probes; constants come out exactly; look-alikes that behave differently are refused."""
    from sspslam_amd.builder import _recognise_cleanup, _recognise_gate
    from sspslam_amd import harness as H
    from sspslam_amd.sspspace import grid_factors_from_table
    space = H.make_ssp_space(2, 55)
    S, _ = space.get_sample_pts_and_ssps(20)
    d = S.shape[1]

    def clean_up_fun(x):                          # the reference's closure, two levels deep under the node's lambda
        sims = S @ x
        return S[np.argmax(sims), :]
    node_fn = lambda t, x: clean_up_fun(x)        # noqa: E731
    kind, table, gf = _recognise_cleanup(node_fn, d, d)
    assert kind == "cleanup" and table is not None and np.array_equal(table, S)
    g0 = space.grid_factors(20)
    assert gf is not None and gf["lhs"].shape == g0["lhs"].shape and gf["rhs"].shape == g0["rhs"].shape

    def sims_of(g, x):
        X = g["dft"] @ x
        Xc = X[0::2] + 1j * X[1::2]
        L, R = g["lhs"][:, 0::2] + 1j * g["lhs"][:, 1::2], g["rhs"][:, 0::2] + 1j * g["rhs"][:, 1::2]
        return np.real((np.conj(Xc)[None, :] * L) @ R.T).reshape(-1)
    x = np.random.RandomState(5).randn(d)
    np.testing.assert_allclose(sims_of(gf, x), S @ x, atol=1e-12)
    np.testing.assert_allclose(sims_of(g0, x), S @ x, atol=1e-12)
    assert grid_factors_from_table(np.random.RandomState(0).randn(400, 55)) is None        # not a grid of SSPs
    softer = lambda t, x: S[np.argsort(S @ x)[-2]]                                        # noqa: E731 - second best row
    assert _recognise_cleanup(softer, d, d) is None

    for thres, rate in ((0.2, 0.2), (0.35, 0.1), (-0.5, 1.5)):
        def update_state_func(t, x, thres=thres, rate=rate):
            if np.allclose(x[-1], 0, atol=1e-3) & (np.sum(x[:d] * x[d:-1]) > thres):
                return rate * (x[:d] - x[d:-1])
            return np.zeros(d)
        assert _recognise_gate(update_state_func, 2 * d + 1, d) == ("gate", d, thres, rate)
    leaky = lambda t, x: 0.2 * (x[:d] - x[d:-1]) if np.sum(x[:d] * x[d:-1]) > 0.2 else 0.01 * x[:d]      # noqa: E731
    assert _recognise_gate(leaky, 2 * d + 1, d) is None
    with nengo.Network(seed=0) as m:
        u = nengo.Node(lambda t: np.ones(d))
        f = nengo.Node(lambda t, x: np.tanh(x), size_in=d, size_out=d)
        nengo.Connection(u, f, synapse=None)
        nengo.Probe(f)
    with pytest.raises(nengo.BuildError, match="function nodes with inputs"):
        build(m)
