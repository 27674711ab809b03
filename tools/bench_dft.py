"""Time of one circular-convolution transform on the device: the four-step MFMA engine vs the Stockham passes vs the dense
matrix, per length.  usage: python tools/bench_dft.py [d ...]     (default 55 217 1015 1801 2049)

A CircularConvolution network stepped in f32 under ssn_run_steps(profile = 2) on the per-operator plan (flag 2097152: one launch
per operator, HIP event pairs): k_dft / k_matvec launches and their average device time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sspslam_amd.frontend as nengo
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.networks import CircularConvolution
from sspslam_amd.simulator import Simulator

PER_OP, FOURSTEP, BIG, MATRIX = 2097152, 536870912, 268435456, 512
for d in [int(x) for x in sys.argv[1:]] or [55, 217, 1015, 1801, 2049]:
    rng = np.random.RandomState(d)
    fa = rng.randn(d) / np.sqrt(d)
    with nengo.Network(seed=1) as m:
        ua = nengo.Node(lambda t, f=fa: f * np.cos(9 * t))
        ea, eb = nengo.Ensemble(60, d), nengo.Ensemble(60, d)
        nengo.Connection(ua, ea, synapse=None)
        nengo.Connection(ua, eb, synapse=None)
        cc = CircularConvolution(12, d)
        nengo.Connection(ea, cc.input_a, synapse=0.005)
        nengo.Connection(eb, cc.input_b, synapse=0.005)
        sink = nengo.Ensemble(30, d)
        nengo.Connection(cc.output, sink, synapse=None)
        nengo.Probe(sink, synapse=0.01)
    model = build(m)
    for name, fl in (("four-step MFMA", BIG | FOURSTEP), ("Stockham", BIG), ("dense matrix", MATRIX)):
        with Simulator(None, model=model, dtype="f32", flags=PER_OP | fl) as sim:
            sim.prepare(400)
            sim.run_steps(100, collect=False)
            sim.run_steps(200, profile=2, collect=False)
            kt = sim.kernel_times()
            c = sim.counters()
            sel = {k: "%d x %.2f us" % (n // 200, 1e3 * ms / n) for k, (n, ms) in kt.items() if k in ("k_dft", "k_matvec")}
            print("d %5d  %-15s fft_transforms %d bluestein %d  %s" % (d, name, c["fft_transforms"], c["fft_bluestein"], sel), flush=True)
