"""ComboLoss of the reference trainer (``framework/Model_Trainer.py:9-23``): the loss whose backward
seeds the hot path's backward.  Plain torch ops on the device (negligible cost)."""
import torch
from torch import nn


class ComboLoss(nn.Module):
    """mean binary cross-entropy + Dice loss per sample, averaged over the batch."""

    def __init__(self):
        super().__init__()
        self.binary_crossentropy = nn.BCELoss(reduction='mean')

    def forward(self, y_pred: torch.Tensor, y_true: torch.Tensor):
        return self.binary_crossentropy(y_pred, y_true) + self.dice_loss(y_pred, y_true)

    @staticmethod
    def dice_loss(y_pred: torch.Tensor, y_true: torch.Tensor):
        batch = y_pred.shape[0]
        overlap = (y_pred * y_true).reshape(batch, -1).sum(-1)
        mass = (y_pred + y_true).reshape(batch, -1).sum(-1)
        return (1 - 2 * overlap / mass).mean()
