"""Loss heads on the HIP kernels (csrc/loss.hip), as autograd Functions.

seg_loss            fused single-pass  BCEWithLogits + Dice (n_classes == 1)  or  CrossEntropy + multiclass
                    Dice (n_classes > 1): the per-step loss of running_files/train_end2end_jsrt.py:181-183.
seg_loss_jaccard    BCEWithLogits + per-sample Jaccard loss of the ISIC variant (train_end2end_isic.py:40-56,247-249).
dice_loss_op        util/dice_score.py:25-28 (one global sum over the batch).
mean_loss           GANLoss (networks.py:263-281), L1Loss and BCEWithLogits (train_end2end_jsrt.py:136-138).
"""
from __future__ import annotations

import torch

from . import ops


def _ws(dev):
    return torch.empty(ops.LOSS_WS, dtype=torch.float32, device=dev)


DICE_EPS = 1e-6


def apply_global_dice(out: torch.Tensor, global_sums: torch.Tensor, world: int) -> None:
    """Turn the per-rank result of gs_seg_loss_fwd into the EXACT global-batch Dice of the reference's bs = world*B
    step (dice_score.py:10-17 sums over the whole batch): `global_sums` = (2*sum p t, sum p, sum t) summed over the
    ranks.  Rewrites out[0], out[2..5] in place and sets the Dice-gradient multiplier out[6] = world, so that the
    gradient AVERAGE the data-parallel exchange computes equals the gradient of the global loss."""
    inter, sp, st = global_sums[0], global_sums[1], global_sums[2]
    sets = sp + st
    sets = torch.where(sets == 0, inter, sets)
    dice_loss = 1.0 - (inter + DICE_EPS) / (sets + DICE_EPS)
    out[2] = dice_loss
    out[0] = out[1] + dice_loss
    out[3:6] = global_sums
    out[6] = float(world)


class _SegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, mask_u8, group):
        logits = logits.contiguous().float()
        out = torch.empty(8, dtype=torch.float32, device=logits.device)
        ops.seg_loss_fwd(logits, mask_u8, _ws(logits.device), out)
        if group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(None if group is True else group)
            if world > 1:
                sums = out[3:6].clone()
                dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=None if group is True else group)   # 12 bytes
                apply_global_dice(out, sums, world)
        ctx.save_for_backward(logits, mask_u8, out)
        return out[0].clone(), out.detach()

    @staticmethod
    def backward(ctx, gout, _gparts):
        logits, mask_u8, out = ctx.saved_tensors
        d = torch.empty_like(logits)
        ops.seg_loss_bwd(logits, mask_u8, out, gout.contiguous().float().reshape(1), 1.0, d)
        return d, None, None


def _mask_u8(mask: torch.Tensor, n: int, h: int, w: int) -> torch.Tensor:
    if mask.dim() == 4:
        if mask.shape[1] != 1:
            raise ValueError("mask must be [N,1,H,W] or [N,H,W] class indices")
        mask = mask[:, 0]
    if tuple(mask.shape) != (n, h, w):
        raise ValueError(f"mask shape {tuple(mask.shape)} does not match logits [{n},*,{h},{w}]")
    return mask.to(torch.uint8).contiguous()


def seg_loss(logits: torch.Tensor, mask: torch.Tensor, return_parts: bool = False, global_dice=None):
    """loss = criterion(logits, mask) + dice_loss(prob(logits), mask) in one pass over the logits.

    logits fp32 [N,C,H,W]; mask integer class indices ([N,1,H,W] or [N,H,W]; {0,1} when C == 1).
    parts = [loss, ce_or_bce, dice_loss, 2*sum(p t), sum p, sum t, dice-grad multiplier, -] (device tensor).
    global_dice: None = per-rank Dice (DDP semantics, the default); True or a process group = the three Dice sums are
    all-reduced (12 bytes) so that loss and averaged gradient are exactly those of the reference's global batch
    (SURVEY 8e); the CE/BCE mean needs no exchange (mean of equal-sized rank means)."""
    n, c, h, w = logits.shape
    loss, parts = _SegLoss.apply(logits, _mask_u8(mask, n, h, w), global_dice)
    return (loss, parts) if return_parts else loss


class _JaccardSegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, mask_u8):
        logits = logits.contiguous().float()
        out = torch.empty(4 + 2 * logits.shape[0], dtype=torch.float32, device=logits.device)
        ops.jaccard_seg_loss_fwd(logits, mask_u8, out)
        ctx.save_for_backward(logits, mask_u8, out)
        return out[0].clone(), out.detach()

    @staticmethod
    def backward(ctx, gout, _gparts):
        logits, mask_u8, out = ctx.saved_tensors
        d = torch.empty_like(logits)
        ops.jaccard_seg_loss_bwd(logits, mask_u8, out, gout.contiguous().float().reshape(1), 1.0, d)
        return d, None


def seg_loss_jaccard(logits: torch.Tensor, mask: torch.Tensor, return_parts: bool = False):
    """The per-step loss of the ISIC / RGB variant (running_files/train_end2end_isic.py:247-249, one class):
    BCEWithLogits(logits, mask) + jaccard_index_loss(sigmoid(logits), mask) (:40-56: per-sample Jaccard, mean over the
    batch, smooth = 1), one fused pass over the logits.  parts = [loss, bce, 1 - mean jaccard, 0, (I_i, S_i)...]."""
    n, c, h, w = logits.shape
    if c != 1:
        raise ValueError("seg_loss_jaccard: the ISIC loss is defined for n_classes == 1")
    loss, parts = _JaccardSegLoss.apply(logits, _mask_u8(mask, n, h, w))
    return (loss, parts) if return_parts else loss


class _DiceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, target):
        inp = inp.contiguous().float()
        target = target.contiguous().float()
        out = torch.empty(4, dtype=torch.float32, device=inp.device)
        ops.dice_loss_fwd(inp, target, _ws(inp.device), out)
        ctx.save_for_backward(target, out)
        ctx.shape = inp.shape
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        target, out = ctx.saved_tensors
        d = torch.empty(ctx.shape, dtype=torch.float32, device=target.device)
        ops.dice_loss_bwd(target, out, gout.contiguous().float().reshape(1), d)
        return d, None


def dice_coeff_per_item(inp: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Mean over items of the per-item Dice coefficient (dice_score.py:5-17, reduce_batch_first=False): inp / target
    [B, ...] with every trailing dim summed per item.  One launch pair for the whole batch; no gradient (validation metric)."""
    if inp.shape != target.shape or inp.dim() < 2:
        raise ValueError("dice_coeff_per_item: inp / target must share a shape [B, ...]")
    B = inp.shape[0]
    p = inp.detach().reshape(B, -1).contiguous().float()
    t = target.detach().reshape(B, -1).contiguous().float()
    out = torch.empty(1 + B, dtype=torch.float32, device=p.device)
    ops.dice_coeff_batched(p, t, out)
    return out[0]


def eval_dice(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Validation Dice of one batch as unet/evaluate.py:29-43 computes it -- (sigmoid > 0.5 | arg-max) prediction, per-sample
    Dice of the foreground class(es), mean -- fused into one pass over the logits."""
    if logits.dim() != 4:
        raise ValueError("eval_dice: logits must be [N,C,H,W]")
    N, C, H, W = logits.shape
    m = _mask_u8(mask, N, H, W)
    out = torch.empty(1 + N * max(1, C - 1), dtype=torch.float32, device=logits.device)
    ops.eval_dice(logits.detach().contiguous().float(), m, out)
    return out[0]


def eval_jaccard(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Validation Jaccard index of one batch as the ISIC script computes it (train_end2end_isic.py:58-84): sigmoid > 0.5
    prediction, per-sample (I + 1) / (P + T - I + 1), mean -- one fused pass over the logits."""
    if logits.dim() != 4 or logits.shape[1] != 1:
        raise ValueError("eval_jaccard: logits must be [N,1,H,W]")
    N, _, H, W = logits.shape
    out = torch.empty(1 + N, dtype=torch.float32, device=logits.device)
    ops.eval_jaccard(logits.detach().contiguous().float(), _mask_u8(mask, N, H, W), out)
    return out[0]


def dice_loss_op(inp: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """1 - dice with ONE global sum over every element (dice_score.py:25-28, reduce_batch_first=True)."""
    return _DiceLoss.apply(inp, target)


class _MeanLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, cval, mode):
        x = x.contiguous().float()
        t = None if t is None else t.contiguous().float()
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        ops.mean_loss_fwd(x, t, cval, mode, _ws(x.device), out)
        ctx.save_for_backward(x, t)
        ctx.cval, ctx.mode = cval, mode
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        x, t = ctx.saved_tensors
        d = torch.empty_like(x)
        ops.mean_loss_bwd(x, t, ctx.cval, ctx.mode, gout.contiguous().float().reshape(1), 1.0, d)
        return d, None, None, None


MODE_BCE_CONST, MODE_MSE_CONST, MODE_MEAN, MODE_L1, MODE_BCE = 0, 1, 2, 3, 4


def mean_loss(x, t=None, cval=0.0, mode=MODE_BCE):
    if t is not None and t.shape != x.shape:
        raise ValueError("input / target shape mismatch")
    return _MeanLoss.apply(x, t, float(cval), int(mode))


def bce_with_logits(x, t):
    """nn.BCEWithLogitsLoss() (mean)."""
    return mean_loss(x, t, 0.0, MODE_BCE)


def l1_loss(a, b):
    """torch.nn.L1Loss() (mean); gradient flows to `a`."""
    return mean_loss(a, b.detach(), 0.0, MODE_L1)
