"""Validation Dice (reference: unet/evaluate.py:10-43): eval-mode forward on the HIP engine (BatchNorm from
running statistics), threshold / arg-max, per-sample Dice (`reduce_batch_first=False`) averaged over batches.
`amp` is accepted for signature parity; the engine already computes in 16-bit with fp32 accumulation."""
import torch
import torch.nn.functional as F

from ..util.dice_score import dice_coeff, multiclass_dice_coeff


@torch.no_grad()
def evaluate(net, dataloader, device, amp=False):
    net.eval()
    num_val_batches = len(dataloader)
    dice_score = 0
    for i, batch in enumerate(dataloader):
        image, mask_true = batch['image'], batch['mask']
        image = image.to(device=device, dtype=torch.float32)
        mask_true = mask_true.to(device=device, dtype=torch.long)
        mask_pred = net(image)
        if net.n_classes == 1:
            assert mask_true.min() >= 0 and mask_true.max() <= 1, 'True mask indices should be in [0, 1]'
            mask_pred = (torch.sigmoid(mask_pred) > 0.5).float()
            dice_score += dice_coeff(mask_pred.squeeze(), mask_true.float().squeeze(), reduce_batch_first=False)
        else:
            assert mask_true.min() >= 0 and mask_true.max() < net.n_classes, \
                'True mask indices should be in [0, n_classes['
            mask_true = F.one_hot(mask_true.squeeze(1), net.n_classes).permute(0, 3, 1, 2).float()
            mask_pred = F.one_hot(mask_pred.argmax(dim=1), net.n_classes).permute(0, 3, 1, 2).float()
            dice_score += multiclass_dice_coeff(mask_pred[:, 1:], mask_true[:, 1:], reduce_batch_first=False)
    net.train()
    return dice_score / max(num_val_batches, 1)
