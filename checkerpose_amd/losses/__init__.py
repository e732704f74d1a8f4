"""Loss head of train.py:307-320 on the device -- same classes as the reference's losses/."""
from .code_loss import MaskedCodeLoss, UnmaskedCodeLoss  # noqa: F401
from .mask_loss import MaskLoss_interpolate  # noqa: F401
