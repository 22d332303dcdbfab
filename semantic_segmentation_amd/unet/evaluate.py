"""Validation Dice on the HIP engine (what unet/evaluate.py:10-43 of the reference returns).

Per batch: eval-mode forward (BatchNorm from the running statistics, folded into the convolutions), then ONE fused pass
over the logits -- prediction (sigmoid > 0.5 for one class, arg-max otherwise), per-sample Dice of the foreground
class(es) (`reduce_batch_first=False`), mean -- instead of the reference's one-hot / permute / per-sample reductions.  The
batch scores are accumulated on the device; the result is their mean over the loader, as a 0-d tensor like the reference.
`amp` is accepted for signature parity: the engine already computes in 16 bits with fp32 accumulation."""
import torch

from ..losses import eval_dice, eval_jaccard


def _evaluate(net, dataloader, device, metric):
    net.eval()
    total = None
    for batch in dataloader:
        logits = net(batch["image"].to(device=device, dtype=torch.float32))
        mask = batch["mask"].to(device=device)
        if mask.dim() == 4 and mask.shape[1] != 1:
            raise ValueError("evaluate: mask must hold class indices [N,1,H,W] or [N,H,W]")
        score = metric(logits, mask)
        total = score if total is None else total + score
    net.train()                       # the reference always leaves the network in train mode (evaluate.py:42)
    if total is None:
        return 0
    return total / max(len(dataloader), 1)


@torch.no_grad()
def evaluate(net, dataloader, device, amp=False):
    """unet/evaluate.py:10-43: mean over the loader of the per-batch Dice."""
    return _evaluate(net, dataloader, device, eval_dice)


@torch.no_grad()
def evaluate_jaccard(net, dataloader, device, amp=False):
    """The `evaluate` of running_files/train_end2end_isic.py:58-84 (one class): mean over the loader of the per-batch
    Jaccard index of the thresholded prediction."""
    return _evaluate(net, dataloader, device, eval_jaccard)
