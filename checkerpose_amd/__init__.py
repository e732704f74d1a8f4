"""checkerpose_amd -- MI355X-native forward hot path of CheckerPose.

    from checkerpose_amd.model.init import InitNet_GNN
    from checkerpose_amd.model.pipeline import PoseNet_GNNskip

are drop-ins for the reference's `model.init` / `model.pipeline` classes (same constructor, forward and
state-dict keys); the arithmetic runs in libcheckerpose_hip.so (include/checkerpose_hip.h).
"""
__version__ = "0.1.0"


def set_deterministic(on=True):
    """Deterministic training mode (also: environment CHECKERPOSE_AMD_DETERMINISTIC=1, or `net.deterministic = True`): the training
    program accumulates BatchNorm sums, weight gradients and Index2Feat's scatter in a fixed order instead of with floating-point
    atomics -- two runs of the same steps give bit-identical parameters, as the reference's CPU step (train.py:303-320) does.
    The switch is process-wide (the library reads it when a launch program is built): models drop their training programs when it
    changes through their own `deterministic` attribute; after calling THIS function directly, call `net.invalidate()`."""
    from . import _abi
    _abi.load().cp_set_deterministic(1 if on else 0)


def is_deterministic():
    from . import _abi
    return bool(_abi.load().cp_get_deterministic())
