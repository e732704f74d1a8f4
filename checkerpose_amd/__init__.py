"""checkerpose_amd -- MI355X-native forward hot path of CheckerPose.

    from checkerpose_amd.model.init import InitNet_GNN
    from checkerpose_amd.model.pipeline import PoseNet_GNNskip

are drop-ins for the reference's `model.init` / `model.pipeline` classes (same constructor, forward and
state-dict keys); the arithmetic runs in libcheckerpose_hip.so (include/checkerpose_hip.h).
"""
__version__ = "0.1.0"
