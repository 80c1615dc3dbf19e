"""torchrl_amd -- MI355X (gfx950) native value estimation for TorchRL-style RL training.

Drop-in for the value-estimation hot path of TorchRL (``torchrl.objectives.value``): the
functional estimators (GAE, TD(lambda), V-trace) and the ``GAE`` / ``TDLambdaEstimator`` /
``VTrace`` modules, backed by hand-written HIP kernels behind a C ABI (include/trl_value.h).
"""
from torchrl_amd._extension import _init_extension

__version__ = "0.1.0"

_init_extension()  # warns (never raises) if the HIP library is absent, as torchrl/__init__.py:22,35
