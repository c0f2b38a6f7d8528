"""`DiscreteVAE.decode` drop-in (/root/reference/indextts/vqvae/xtts_dvae.py:332-351) on the HIP engine."""
from __future__ import annotations

import numpy as np
import torch


class DiscreteVAE:
    def __init__(self, engine):
        self._eng = engine

    @torch.no_grad()
    def decode(self, img_seq):
        """codes [B, T] -> (mel [B, channels, 4T], None): the reference also returns the penultimate activation,
        which the engine does not materialise."""
        codes = img_seq.detach().cpu().numpy() if isinstance(img_seq, torch.Tensor) else np.asarray(img_seq)
        return self._eng.dvae_decode(codes), None
