"""The torch-CPU formulation of the generator lives in oracle/ae_torch.py (it is also the CPU-baseline network of bench.py)."""
from oracle.ae_torch import _conv, _deconv, forward  # noqa: F401
