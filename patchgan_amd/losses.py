"""Loss functions with the reference's names and argument order (patchgan/losses.py:5-39) for code that calls them
directly.  Inside ``Trainer.batch`` the losses and their gradients run as HIP kernels (engine.loss_value_and_grad);
these torch-op versions exist for API compatibility and work on any device torch supports."""
import torch
from torch import nn


def tversky(y_true, y_pred, beta, batch_mean=True):
    tp = torch.sum(y_true * y_pred, dim=(1, 2, 3))
    fn = torch.sum((1. - y_pred) * y_true, dim=(1, 2, 3))
    fp = torch.sum(y_pred * (1. - y_true), dim=(1, 2, 3))
    t = tp / (tp + beta * fn + (1. - beta) * fp)
    return torch.mean(1. - t) if batch_mean else (1. - t)


def fc_tversky(y_true, y_pred, beta, gamma=0.75, batch_mean=True):
    tp = torch.sum(y_true * y_pred, dim=(1, 2, 3))
    fn = torch.sum((1. - y_pred) * y_true, dim=(1, 2, 3))
    fp = torch.sum(y_pred * (1. - y_true), dim=(1, 2, 3))
    t = (tp + 1) / (tp + beta * fn + (1. - beta) * fp + 1)
    ft = 1 - t
    return torch.pow(torch.mean(ft), gamma) if batch_mean else torch.pow(ft, gamma)


def MAE_loss(y_true, y_pred):
    return torch.mean(torch.abs(y_true - y_pred))


bce_loss = nn.BCELoss()
