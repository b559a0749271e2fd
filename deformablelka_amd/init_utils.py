"""Initialisation helpers for synthetic-data benchmarks."""
import torch


def randomize_offset_nets(module, std=0.02, seed=123):
    """Fresh D-LKA modules predict zero offsets (zero-initialised conv_offset, 3D/dcn/modules/deform_conv.py:86-88; the 2-D offset_net is a
    default-initialised conv) — give the offset predictors small random weights so that samples are not integer-aligned."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "conv_offset" in name or "offset_net" in name:
                p.copy_((torch.randn(p.shape, generator=gen) * std).to(p.device))
