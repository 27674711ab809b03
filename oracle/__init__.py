"""ORACLE - test infrastructure only (see oracle/stepper.py header).  Never imported by the product."""
from .stepper import OracleSimulator, lif_step, lif_rate  # noqa: F401
