"""Counterpart of the reference's transformer/utils.py:1-9.  The reference keeps a global
DEVICE set by init_device(); here every op takes the device from its tensors, DEVICE is
kept only so that scripts written against the reference keep working."""
import torch

DEVICE = None


def init_device():
    global DEVICE
    DEVICE = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    return DEVICE


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
