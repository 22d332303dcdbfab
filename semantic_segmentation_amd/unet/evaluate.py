"""Validation Dice on the HIP engine (what unet/evaluate.py:10-43 of the reference returns).

Per batch: eval-mode forward (BatchNorm from the running statistics, folded into the convolutions), then ONE fused pass
over the logits -- prediction (sigmoid > 0.5 for one class, arg-max otherwise), per-sample Dice of the foreground
class(es) (`reduce_batch_first=False`), mean -- instead of the reference's one-hot / permute / per-sample reductions.  The
batch scores are accumulated on the device; the result is their mean over the loader, as a 0-d tensor like the reference.
`amp` is accepted for signature parity: the engine already computes in 16 bits with fp32 accumulation."""
import torch

from ..losses import eval_dice


@torch.no_grad()
def evaluate(net, dataloader, device, amp=False):
    net.eval()
    total, batches = None, 0
    for batch in dataloader:
        logits = net(batch["image"].to(device=device, dtype=torch.float32))
        mask = batch["mask"].to(device=device)
        if mask.dim() == 4 and mask.shape[1] != 1:
            raise ValueError("evaluate: mask must hold class indices [N,1,H,W] or [N,H,W]")
        score = eval_dice(logits, mask)
        total = score if total is None else total + score
        batches += 1
    net.train()                       # the reference always leaves the network in train mode (evaluate.py:42)
    if total is None:
        return 0
    return total / max(len(dataloader), 1)
