"""Checkpoint reader: what `load_checkpoint(model, path)` consumes (/root/reference/indextts/utils/checkpoint.py:25-34):
a torch pickle, optionally nested under "model"; returns numpy fp32 state dict for the weight packer."""
from __future__ import annotations

import numpy as np
import torch


def read_state_dict(path: str, key: str = None):
    # tensors only, as the reference's torch.load default (weights_only=True on current torch): a downloaded checkpoint
    # must not be able to run pickled code.  ITTS_UNSAFE_PICKLE=1 is the explicit opt-in for legacy pickles.
    import os

    ck = torch.load(path, map_location="cpu", weights_only=os.environ.get("ITTS_UNSAFE_PICKLE") != "1")
    if key is not None and key in ck:
        ck = ck[key]
    elif "model" in ck and isinstance(ck["model"], dict):
        ck = ck["model"]
    out = {}
    for k, v in ck.items():
        if isinstance(v, torch.Tensor) and not k.startswith("inference_model."):
            out[k] = v.detach().float().numpy() if v.dtype.is_floating_point else v.numpy()
    return out
