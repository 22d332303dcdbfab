"""Dice coefficient / loss with the reference's signatures (util/dice_score.py:5-28) on the HIP
reduction kernels.  Results follow the reference: `sum_dim` selection (:10), inter = 2*sum(p t) (:12),
sets = sum p + sum t with the exact-zero replacement (:13-14), eps 1e-6 (:16), mean over the kept dims (:17)."""
import torch
from torch import Tensor

from ..losses import dice_coeff_per_item, dice_loss_op


def dice_coeff(input: Tensor, target: Tensor, reduce_batch_first: bool = False, epsilon: float = 1e-6):
    assert input.size() == target.size()
    assert input.dim() == 3 or not reduce_batch_first
    if epsilon != 1e-6:
        raise NotImplementedError("the HIP dice kernels are built for the reference's epsilon = 1e-6")
    if input.dim() == 2 or reduce_batch_first:
        return 1 - dice_loss_op(input, target)          # one global reduction
    # per-item sums over the last two dims, then the mean over the leading ones (dice_score.py:10,17)
    items_in, items_t = input.reshape(-1, *input.shape[-2:]), target.reshape(-1, *target.shape[-2:])
    if torch.is_grad_enabled() and (input.requires_grad or target.requires_grad):
        # differentiable like the reference (the batched kernel below has no backward): one global-reduction op per item.
        # The reference only calls this branch from evaluate.py under no_grad; a loss built on it still gets its gradient.
        vals = [1 - dice_loss_op(a, b) for a, b in zip(items_in, items_t)]
        return torch.stack(vals).mean()
    # validation metric: one launch pair for the whole batch
    return dice_coeff_per_item(items_in, items_t)


def multiclass_dice_coeff(input: Tensor, target: Tensor, reduce_batch_first: bool = False, epsilon: float = 1e-6):
    return dice_coeff(input.flatten(0, 1), target.flatten(0, 1), reduce_batch_first, epsilon)


def dice_loss(input: Tensor, target: Tensor, multiclass: bool = False):
    fn = multiclass_dice_coeff if multiclass else dice_coeff
    return 1 - fn(input, target, reduce_batch_first=True)
