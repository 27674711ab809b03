"""Importable alias for the package directory ``semantic-spiking-neural-slam-2023_amd/``.

The layout contract names the package directory with hyphens, which the ``import`` statement
cannot spell.  This alias package has no code of its own: it points ``__path__`` at the real
directory and executes that directory's ``__init__.py`` in this namespace, so
``import sspslam_amd.simulator`` loads ``semantic-spiking-neural-slam-2023_amd/simulator.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "semantic-spiking-neural-slam-2023_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
