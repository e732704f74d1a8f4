"""Drop-in for the reference's losses/code_loss.py (UnmaskedCodeLoss :6-27, MaskedCodeLoss :30-62): same
constructor argument, forward signature and value; value + gradient come from one fused HIP launch (cp_code_loss)."""
import torch.nn as nn

from .._abi import LOSS_BCE, LOSS_L1
from ._fn import _CeLossFn, _CodeLossFn

_TYPES = {"BCE": LOSS_BCE, "L1": LOSS_L1}


class UnmaskedCodeLoss(nn.Module):
    def __init__(self, loss_type="BCE"):
        super().__init__()
        if loss_type not in _TYPES:
            raise ValueError("loss_type {} not supported in MaskedCodeLoss".format(loss_type))   # reference's message
        self.loss_type = loss_type

    def forward(self, pred_code_prob, gt_code):
        """pred_code_prob, gt_code: (batch, #bits, #keypoints)"""
        return _CodeLossFn.apply(pred_code_prob, gt_code, None, _TYPES[self.loss_type])


class MaskedCodeLoss(nn.Module):
    def __init__(self, loss_type="BCE"):
        super().__init__()
        if loss_type != "CE" and loss_type not in _TYPES:     # "CE": multi-class (code_loss.py:36-37), cp_masked_ce_loss
            raise ValueError("loss_type {} not supported in MaskedCodeLoss".format(loss_type))
        self.loss_type = loss_type

    def forward(self, pred_code_prob, gt_code, gt_mask):
        """pred_code_prob, gt_code: (batch, #bits, #keypoints) -- gt_code (batch, 1, #keypoints) class ids for "CE";
        gt_mask: (batch, 1, #keypoints)"""
        if self.loss_type == "CE":
            return _CeLossFn.apply(pred_code_prob, gt_code, gt_mask)
        return _CodeLossFn.apply(pred_code_prob, gt_code, gt_mask, _TYPES[self.loss_type])
