"""Independent check of the lowering (parity item 1): ``oracle/graphwalk.py`` steps the front-end object
graph directly - one Python object per Node / Ensemble / Connection, nengo's sets -> incs -> reads -> updates order,
one dt of delay per synapse - and never sees the operator list ``builder.py`` lowers for the GPU and for
``oracle/stepper.py``.  Both must agree on probes, learned PES decoders and Voja-moved encoders."""
import numpy as np
import pytest

import sspslam_amd.frontend as nengo
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from oracle import OracleSimulator
from oracle.graphwalk import GraphWalkSimulator

from helpers import random_network, small_pathint


def _probe_index(model, probe):
    return [i for i, p in enumerate(model.probes) if p["probe"] is probe][0]


@pytest.mark.parametrize("staged", [True, False])
def test_pathintegration_lowering_equals_the_graph_walk(staged):
    """PathIntegration d = 55, n = 60 (reference networks/pathintegration.py:162-191): EnsembleArray slices, per-VCO
    recurrent connections through Lowpass(tau), default-synapse read-in / read-out, filtered probe."""
    pm = small_pathint(ssp_dim=55, n=60, T=10.0, limit=0.2)
    with pm.model:
        p_osc = nengo.Probe(pm.pathintegrator.oscillators.output[3:9], synapse=None)
    model = build(pm.model, staged=staged)
    ref = OracleSimulator(model)
    walk = GraphWalkSimulator(pm.model, model)
    ref.run_steps(300)
    walk.run_steps(300)
    for p in (pm.probe, p_osc):
        a, b = ref.probe_data(_probe_index(model, p)), walk.probe_data(p)
        assert a.shape == b.shape and np.abs(b).max() > 0.05
        np.testing.assert_allclose(a, b, atol=1e-9, rtol=0)


def test_slam_lowering_equals_the_graph_walk():
    """Small SLAMNetwork (reference slam.py:241-307, associativememory.py:30-54): gate and clean-up function nodes
    called as Python, circular convolutions, direct neuron inhibition, PES and Voja with their one-step-late
    `target += delta`, weight / encoder / neuron probes."""
    s = H.make_ssp_space(2, 55)
    path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
    sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons=120, circonv_n_neurons=30,
                           view_rad=0.6, weights_sample_every=0.05)
    am = sm.slam.assomemory
    with sm.model:
        p_enc = nengo.Probe(am.conn_in.learning_rule, "scaled_encoders", sample_every=0.05)
        p_recall = nengo.Probe(am.recall, synapse=0.05)
        p_spk = nengo.Probe(am.memory.neurons[:40])
        p_clean = nengo.Probe(sm.slam.gridcells, synapse=None)
    model = build(sm.model)
    ref = OracleSimulator(model)
    walk = GraphWalkSimulator(sm.model, model)
    n = 400
    ref.run_steps(n)
    walk.run_steps(n)
    for p in (sm.probe, p_recall, p_spk, p_clean, sm.weights_probe, p_enc):
        a, b = ref.probe_data(_probe_index(model, p)), walk.probe_data(p)
        assert a.shape == b.shape, (p, a.shape, b.shape)
        np.testing.assert_allclose(a, b, atol=1e-9, rtol=0, err_msg=repr(p))
    # something was learned, the encoders moved, and the final state agrees
    W = ref.buf[model.params[am.conn_out].learned_buffer]
    E = ref.buf[model.params[am.memory].encoder_buffer]
    assert np.abs(W).max() > 0 and np.abs(E - model.params[am.memory].scaled_encoders).max() > 0
    # (the walker holds the delta of the last step back, as nengo does: add it before comparing final states)
    rules = {r.kind: r for r in walk.rules}
    np.testing.assert_allclose(walk.weights(am.conn_out) + rules["PES"].delta, W, atol=1e-12, rtol=0)
    np.testing.assert_allclose(walk.scaled_encoders(am.memory) + rules["Voja"].delta, E, atol=1e-10, rtol=0)


def test_weight_probe_samples_exclude_the_delta_of_their_own_step():
    """nengo applies `weights += delta` as an inc at the start of the NEXT step: the first sample of a weight probe
    taken every step is the initial matrix."""
    with nengo.Network(seed=3) as net:
        stim = nengo.Node(lambda t: [0.8])
        pre = nengo.Ensemble(40, 1)
        post = nengo.Node(size_in=1)
        err = nengo.Node(lambda t: [1.0])
        nengo.Connection(stim, pre, synapse=None)
        c = nengo.Connection(pre, post, function=lambda x: [0.0], learning_rule_type=nengo.PES(1e-3), synapse=None)
        nengo.Connection(err, c.learning_rule, synapse=None)
        pw = nengo.Probe(c, "weights")
        nengo.Probe(post)          # (a connection into a node nothing reads would be dropped as a dead end)
    model = build(net)
    ref = OracleSimulator(model)
    walk = GraphWalkSimulator(net, model)
    ref.run_steps(60)
    walk.run_steps(60)
    a, b = ref.probe_data(_probe_index(model, pw)), walk.probe_data(pw)
    np.testing.assert_allclose(a, b, atol=1e-15, rtol=0)
    np.testing.assert_array_equal(a[0], model.buffers[model.params[c].learned_buffer])
    assert np.abs(a[-1] - a[0]).max() > 0


def test_gridcell_population_options_lower_like_the_graph_walk():
    """with_gcs=True (reference pathintegration.py:150-154: the output is an ensemble with grid-cell encoders) and
    gc_n_neurons > 0 (slam.py:274-281: clean-up node -> grid-cell ensemble -> binding network through Lowpass(tau))."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("gpu_parity_helpers", os.path.join(os.path.dirname(__file__), "test_gpu_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for net, probes in mod._gridcell_models():
        model = build(net)
        assert any(o["kind"] == "neurons" for o in model.ops)
        ref = OracleSimulator(model)
        walk = GraphWalkSimulator(net, model)
        ref.run_steps(250)
        walk.run_steps(250)
        for p in probes:
            a, b = ref.probe_data(_probe_index(model, p)), walk.probe_data(p)
            assert a.shape == b.shape and np.abs(a).max() > 0
            np.testing.assert_allclose(a, b, atol=1e-9, rtol=0)


@pytest.mark.parametrize("seed", [2, 5, 6, 12, 17, 20, 22, 23])
def test_random_networks_lowering_equals_the_graph_walk(seed):
    """Seeded random networks (helpers.random_network: nodes, ensembles of every neuron type, an ensemble array, pass-through
    nodes, decoded / direct / recurrent connections, probes on everything; the same generator feeds the GPU fuzz tests):
    the lowered operator list - merged, pruned, partitioned into stages - against the object-graph interpreter."""
    net, probes = random_network(seed, learned_probes=True)      # (+ the learned decoders / encoders where the seed has a rule)
    model = build(net)
    ref = OracleSimulator(model)
    walk = GraphWalkSimulator(net, model)
    ref.run_steps(80)
    walk.run_steps(80)
    for p in probes:
        a, b = ref.probe_data(_probe_index(model, p)), walk.probe_data(p)
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, atol=1e-9, rtol=0, err_msg=f"seed {seed}")
