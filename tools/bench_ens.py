"""Micro-benchmark of the ensemble-array kernel at config-2 size with realistic LIF statistics
(nengo-default gains/biases, unit-circle oscillator states -> ~10 % of neurons spike per step) and
random decoders (no solver).  usage: bench_ens.py [K] [n] [steps] [flags...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd.builder import BuiltModel
from sspslam_amd.simulator import Simulator
import sspslam_amd.frontend as fe

K = int(sys.argv[1]) if len(sys.argv) > 1 else 508
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
flag_list = [int(f) for f in sys.argv[4:]] or [0, 2]
rng = np.random.RandomState(0)
lif = fe.LIF()
enc = rng.randn(K, n, 3); enc /= np.linalg.norm(enc, axis=2, keepdims=True)
gain, bias = lif.gain_bias(rng.uniform(200, 400, (K, n)), rng.uniform(-1, 0.9, (K, n)))
enc = (enc * (gain / np.sqrt(2))[:, :, None]).transpose(0, 2, 1).copy()
th = rng.uniform(0, 2 * np.pi, K)
x = np.stack([np.cos(th), np.sin(th), rng.uniform(-0.3, 0.3, K)], 1)
m = BuiltModel(0.001)
m.sig_size = 3 * K + 5 * K
m.sig_init = np.zeros(m.sig_size); m.sig_init[:3 * K] = x.reshape(-1)
dec = rng.randn(K, 5, n) * 1e-4
idx = (3 * K + np.arange(5 * K)).reshape(K, 5).astype(np.int32)
b = [m.add_buffer(a, nm, role) for a, nm, role in ((enc, "enc", "param"), (bias, "bias", "param"), (dec, "dec", "param"), (idx, "idx", "index"),
                                                   (np.zeros((K, n)), "v", "state"), (np.zeros((K, n)), "r", "state"))]
nd = dict(type="lif", tau_rc=0.02, tau_ref=0.002, min_voltage=0.0, amplitude=1.0)
m.ops = [dict(kind="ensarray", x=0, K=K, n=n, din=3, dout=5, enc=b[0], bias=b[1], dec=b[2], dst_idx=b[3], v=b[4], r=b[5], neuron=nd,
              level=0, label="big", k_lo=0, k_total=K)]
probe = object()
m.probes = [dict(probe=probe, src=3 * K, width=5 * K, every=1)]
outs = {}
for flags in flag_list:
    sim = Simulator(None, model=m, dtype="f32", flags=flags)
    sim.run_steps(100)                       # reach stationary spiking
    R = sim.read_buffer(b[5])
    sim.run_steps(steps, profile=True, collect=False)
    c = sim.counters()
    us = c["dominant_ms_total"] / c["dominant_launches"] * 1e3
    print("flags %d: k_ensarray avg %.2f us (events) -> %.0f GB/s algorithmic; refractory fraction %.3f; total %.1f us/step" %
          (flags, us, c["dominant_bytes_per_launch"] / us / 1e3, (R > 0).mean(), c["last_run_ms"] / steps * 1e3), flush=True)
    sim._collect()
    outs[flags] = sim.data[probe]
    sim.close()
ks = sorted(outs)
for k in ks[1:]:
    print("flags %d vs %d identical:" % (k, ks[0]), np.array_equal(outs[k], outs[ks[0]]))
