"""`DiscreteVAE` drop-in on the HIP engine: `decode` (/root/reference/indextts/vqvae/xtts_dvae.py:332-351) and
`get_codebook_indices` (:325-330)."""
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
        with self._eng.lock:
            return self._eng.dvae_decode(codes), None

    @torch.no_grad()
    def get_codebook_indices(self, images):
        """mel [B, channels, T] -> codes int64 [B, T / 4] (encoder + nearest codebook row, xtts_dvae.py:86-92,325-330)."""
        with self._eng.lock:
            return torch.from_numpy(self._eng.dvae_encode(images)).to(self._eng.device)
