"""autograd glue for the fused loss kernels: the forward launch computes the loss value AND d loss / d logits
(cp_code_loss / cp_mask_loss); backward only scales the saved gradient by the incoming scalar."""
import torch

from .. import _abi


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("checkerpose_amd.losses: CUDA/HIP tensors required (no CPU fallback)")


def _workspace(dev):
    return torch.empty(_abi.load().cp_loss_workspace_bytes(), dtype=torch.uint8, device=dev)


def _batch_view(t, inner):
    """(ptr-carrying tensor, batch stride in elements) of a (B, ...) fp32 tensor whose trailing dims are contiguous
    (`inner` elements per batch entry); anything else is made contiguous first."""
    if t.dtype != torch.float32:
        t = t.float()
    want, acc = [], 1
    for d in reversed(t.shape[1:]):
        want.append(acc)
        acc *= d
    if list(t.stride()[1:]) != list(reversed(want)) or (t.shape[0] > 1 and t.stride(0) < inner):
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else inner)


class _CodeLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask, loss_type):
        _require_cuda(pred, gt, mask)
        lib = _abi.load()
        B, nb, N = pred.shape
        if gt.shape[0] != B or gt.shape[1] != nb or gt.shape[2] != N:
            raise ValueError("gt_code must be (B, #bits, #keypoints) like the prediction")
        p, pbs = _batch_view(pred.detach(), nb * N)
        g, gbs = _batch_view(gt, nb * N)
        m = None
        if mask is not None:
            if tuple(mask.shape) != (B, 1, N):
                raise ValueError("gt_mask must be (B, 1, #keypoints)")
            m = mask.float().contiguous()
        need = pred.requires_grad
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        dp = torch.empty(B, nb, N, dtype=torch.float32, device=pred.device) if need else None
        ws = _workspace(pred.device)
        st = torch.cuda.current_stream(pred.device).cuda_stream
        _abi.check(lib.cp_code_loss(st, loss_type, p.data_ptr(), pbs, g.data_ptr(), gbs, m.data_ptr() if m is not None else None,
                                    B, nb, N, loss.data_ptr(), dp.data_ptr() if need else None, nb * N, ws.data_ptr()),
                   "cp_code_loss")
        ctx.dp = dp
        return loss

    @staticmethod
    def backward(ctx, gl):
        return (ctx.dp * gl if ctx.dp is not None else None), None, None, None


class _CeLossFn(torch.autograd.Function):
    """MaskedCodeLoss("CE"): pred (B, C, N) class logits, gt (B, 1, N) class ids, mask (B, 1, N) (code_loss.py:47-61)"""

    @staticmethod
    def forward(ctx, pred, gt, mask):
        _require_cuda(pred, gt, mask)
        lib = _abi.load()
        B, Cc, N = pred.shape
        if tuple(gt.shape) != (B, 1, N) or tuple(mask.shape) != (B, 1, N):
            raise ValueError("CE: gt_code and gt_mask must be (B, 1, #keypoints)")
        p, pbs = _batch_view(pred.detach(), Cc * N)
        g = gt[:, 0, :].float().contiguous()
        m = mask.float().contiguous()
        need = pred.requires_grad
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        dp = torch.empty(B, Cc, N, dtype=torch.float32, device=pred.device) if need else None
        ws = _workspace(pred.device)
        st = torch.cuda.current_stream(pred.device).cuda_stream
        _abi.check(lib.cp_masked_ce_loss(st, p.data_ptr(), pbs, g.data_ptr(), m.data_ptr(), B, Cc, N, loss.data_ptr(),
                                         dp.data_ptr() if need else None, Cc * N, ws.data_ptr()), "cp_masked_ce_loss")
        ctx.dp = dp
        return loss

    @staticmethod
    def backward(ctx, gl):
        return (ctx.dp * gl if ctx.dp is not None else None), None, None


class _MaskLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        _require_cuda(pred, gt)
        lib = _abi.load()
        B, Cc, h, w = pred.shape
        if gt.dim() != 3 or gt.shape[0] != B:
            raise ValueError("groundtruth_mask must be (B, H, W)")
        p, pbs = _batch_view(pred.detach(), h * w)       # channel 0 of the passed slice starts at data_ptr()
        g = gt.float().contiguous()
        need = pred.requires_grad
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        dp = torch.zeros(B, Cc, h, w, dtype=torch.float32, device=pred.device) if need else None
        ws = _workspace(pred.device)
        st = torch.cuda.current_stream(pred.device).cuda_stream
        _abi.check(lib.cp_mask_loss(st, p.data_ptr(), pbs, g.data_ptr(), B, h, w, g.shape[1], g.shape[2], loss.data_ptr(),
                                    dp.data_ptr() if need else None, Cc * h * w, ws.data_ptr()), "cp_mask_loss")
        ctx.dp = dp
        return loss

    @staticmethod
    def backward(ctx, gl):
        return (ctx.dp * gl if ctx.dp is not None else None), None
