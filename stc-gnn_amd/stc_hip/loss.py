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
    def _sample_sums(t: torch.Tensor) -> torch.Tensor:
        """Sum over everything but the batch dimension.  A (batch, n) -> (batch,) reduction gives the device one
        workgroup per sample (1.6 ms for 2 x 9.6 M elements on MI355X); two stages keep the whole GPU busy."""
        batch = t.shape[0]
        n = t.numel() // max(batch, 1)
        groups = 1
        while groups < 4096 and n % (2 * groups) == 0 and n // (2 * groups) >= 1024:
            groups *= 2
        flat = t.reshape(batch, groups, -1) if groups > 1 else t.reshape(batch, 1, -1)
        return flat.sum(-1).sum(-1)

    @classmethod
    def dice_loss(cls, y_pred: torch.Tensor, y_true: torch.Tensor):
        overlap = cls._sample_sums(y_pred * y_true)
        mass = cls._sample_sums(y_pred + y_true)
        return (1 - 2 * overlap / mass).mean()
