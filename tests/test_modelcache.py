"""Built-model cache (sspslam_amd/modelcache.py): a hit is interchangeable with a fresh build, and anything that changes
the build changes the key."""
import numpy as np
import pytest

from sspslam_amd import harness as H
from sspslam_amd import modelcache as MC
from sspslam_amd.builder import build
from oracle import OracleSimulator

from helpers import small_pathint


def _same_model(a, b):
    assert a.sig_size == b.sig_size and a.dt == b.dt
    assert np.array_equal(a.sig_init, b.sig_init)
    assert len(a.buffers) == len(b.buffers)
    for x, y in zip(a.buffers, b.buffers):
        assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y)      # bit-identical buffers
    assert len(a.ops) == len(b.ops)
    for o, q in zip(a.ops, b.ops):
        assert set(o) == set(q)
        for k in o:
            assert np.array_equal(np.asarray(o[k], dtype=object) if isinstance(o[k], (list, tuple, dict)) else o[k], q[k]) or o[k] == q[k], k
    assert getattr(a, "stage_info", None) == getattr(b, "stage_info", None)
    assert getattr(a, "exchange", None) == getattr(b, "exchange", None)


def _small_slam(seed=0, view_rad=0.6):
    space = H.make_ssp_space(2, ssp_dim=7)
    path, vels = H.make_random_path(4.0, limit=0.5, seed=3)
    return H.make_slam_model(space, path, vels, n_landmarks=3, pi_n_neurons=40, mem_n_neurons=30, circonv_n_neurons=20,
                             view_rad=view_rad, seed=seed, weights_sample_every=0.01)


def test_hit_equals_fresh_build_and_binds_to_the_live_network(tmp_path, monkeypatch):
    monkeypatch.setenv("SSN_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("SSN_NO_CACHE", raising=False)
    sm1 = _small_slam()
    fresh = build(sm1.model, n_eval_points=300)
    first = MC.cached_build(sm1.model, n_eval_points=300)
    assert first.stats["cache"] == "miss"
    _same_model(fresh, first)
    # a second process: the same declaration made again - other Python objects, same fingerprint
    sm2 = _small_slam()
    hit = MC.cached_build(sm2.model, n_eval_points=300)
    assert hit.stats["cache"] == "hit" and hit.stats["cache_key"] == first.stats["cache_key"]
    _same_model(fresh, hit)
    # the hit is bound to sm2's objects: probes, ensembles, connections and the table functions of ITS nodes
    assert {p["probe"] for p in hit.probes} == set(sm2.model.all_probes)
    assert set(hit.params) >= set(sm2.model.all_ensembles)
    assert not (set(hit.params) & set(sm1.model.all_ensembles))
    for tb in hit.tables:
        assert tb["node"] in sm2.model.all_nodes and tb["fn"] is tb["node"].output
    am = sm2.slam.assomemory
    assert hit.params[am.conn_out].learned_buffer == fresh.params[sm1.slam.assomemory.conn_out].learned_buffer
    # ... and steps like the fresh one (probes of signals AND of learned buffers)
    a, b = OracleSimulator(fresh), OracleSimulator(hit)
    a.run_steps(60)
    b.run_steps(60)
    for i in range(len(fresh.probes)):
        assert np.array_equal(np.asarray(a.probe_data(i)), np.asarray(b.probe_data(i)))


def test_what_changes_the_build_changes_the_key(tmp_path, monkeypatch):
    monkeypatch.setenv("SSN_CACHE_DIR", str(tmp_path))
    base = MC.fingerprint(_small_slam().model, n_eval_points=300)
    assert base == MC.fingerprint(_small_slam().model, n_eval_points=300)
    assert base != MC.fingerprint(_small_slam().model, n_eval_points=301)
    assert base != MC.fingerprint(_small_slam(seed=1).model, n_eval_points=300)
    assert base != MC.fingerprint(_small_slam().model, n_eval_points=300, staged=False)
    assert base != MC.fingerprint(_small_slam().model, n_eval_points=300, neuron_shard=(0, 2))
    # another path is another model (its largest velocity scales the velocity transforms, reference slam.py:392) ...
    pm1 = small_pathint(seed=1)
    assert MC.fingerprint(pm1.model) == MC.fingerprint(small_pathint(seed=1).model)
    assert MC.fingerprint(pm1.model) != MC.fingerprint(small_pathint(seed=2).model)
    # ... and so is another decoder target: the oscillators' feedback closure carries tau (reference pathintegration.py:118-125)
    assert MC.fingerprint(pm1.model) != MC.fingerprint(small_pathint(seed=1, tau=0.06).model)
    # a constant hidden in a function node's closure (the gate's threshold, reference slam.py:233-237)
    space = H.make_ssp_space(2, ssp_dim=7)
    path, vels = H.make_random_path(4.0, limit=0.5, seed=3)
    kw = dict(n_landmarks=3, pi_n_neurons=40, mem_n_neurons=30, circonv_n_neurons=20, view_rad=0.6)
    f1 = MC.fingerprint(H.make_slam_model(space, path, vels, update_thres=0.2, **kw).model, n_eval_points=300)
    f2 = MC.fingerprint(H.make_slam_model(space, path, vels, update_thres=0.25, **kw).model, n_eval_points=300)
    assert f1 != f2


def test_unseeded_or_switched_off_just_builds(tmp_path, monkeypatch):
    monkeypatch.setenv("SSN_CACHE_DIR", str(tmp_path))
    import sspslam_amd.frontend as nengo
    with nengo.Network() as net:          # no seed anywhere: every build samples differently
        e = nengo.Ensemble(20, 1)
        nengo.Probe(e)
    m = MC.cached_build(net)
    assert m.stats["cache"].startswith("uncacheable") and not list(tmp_path.iterdir())
    monkeypatch.setenv("SSN_NO_CACHE", "1")
    assert MC.cached_build(small_pathint().model).stats["cache"] == "off"
    assert not list(tmp_path.iterdir())


def test_a_damaged_entry_is_rebuilt(tmp_path, monkeypatch):
    monkeypatch.setenv("SSN_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("SSN_NO_CACHE", raising=False)
    pm = small_pathint()
    m = MC.cached_build(pm.model)
    (entry,) = list(tmp_path.iterdir())
    entry.write_bytes(entry.read_bytes()[:100])
    again = MC.cached_build(small_pathint().model)
    assert again.stats["cache"].startswith("miss (entry unreadable")
    _same_model(m, again)
    assert MC.cached_build(small_pathint().model).stats["cache"] == "hit"
