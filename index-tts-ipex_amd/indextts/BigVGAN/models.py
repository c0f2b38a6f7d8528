"""`BigVGAN` generator drop-in (/root/reference/indextts/BigVGAN/models.py:130-260) on the HIP engine."""
from __future__ import annotations

import torch


class BigVGAN:
    def __init__(self, engine):
        self._eng = engine
        self._spk_key, self._spk = None, None

    def speaker_encoder(self, mel_ref, lens=None):
        """ECAPA_TDNN.forward: [B, F, n_mels] -> [B, 1, E] (ECAPA_TDNN.py:545-581)."""
        with self._eng.lock:
            return self._eng.ecapa(mel_ref).unsqueeze(1)

    def remove_weight_norm(self):  # folded by the weight packer (models.py:252-260)
        return None

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    @torch.no_grad()
    def forward(self, x, mel_ref, lens=None):
        """latent [B, T, gpt_dim], mel_ref [B, F, n_mels] -> (wav [B, 1, T*1024], None) (models.py:201-250).
        The speaker embedding is cached per prompt tensor (the reference recomputes it per sentence)."""
        with self._eng.lock:
            # keyed on the tensor object (a reference is held: its storage cannot be recycled for another prompt)
            hit = self._spk_key is not None and self._spk_key[0] is mel_ref and self._spk_key[1] == mel_ref._version
            if not hit:
                self._spk, self._spk_key = self._eng.ecapa(mel_ref), (mel_ref, mel_ref._version)
            spk = self._spk
            if spk.shape[0] == 1 and x.shape[0] > 1:
                spk = spk.expand(x.shape[0], -1).contiguous()
            return self._eng.bigvgan(x, spk), None

    __call__ = forward
