"""MI355X-native simulator backend for the SSP-SLAM spiking networks.

Hot path (BASELINE.json north_star): the per-timestep loop that advances
``PathIntegration`` / ``SLAMNetwork`` behind the ``nengo.Simulator`` API, executed by hand-written
HIP kernels for gfx950 in ``csrc/`` behind the C ABI of ``include/ssn.h``.

Import as ``sspslam_amd`` (alias of this hyphen-named directory).  Sub-modules:

* ``sspspace``   SSP algebra (build time / harness)             reference sspslam/sspspace.py
* ``frontend``   nengo-shaped object model (Network, Node, ...)  reference's ``import nengo`` call sites
* ``networks``   PathIntegration, CircularConvolution, ...       reference sspslam/networks/*.py
* ``builder``    Network -> frozen BuiltModel (operator list)
* ``simulator``  ``Simulator(model)`` / ``run`` / ``data[probe]`` over ctypes -> libssn_hip.so
"""
__version__ = "0.1.0"

from .sspspace import SPSpace, SSPSpace, HexagonalSSPSpace, RandomSSPSpace  # noqa: F401
