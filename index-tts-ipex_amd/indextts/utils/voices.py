"""Saved-voice features in the reference web UI's on-disk format (/root/reference/webui.py:56-58,200-221,309-313): a
voice is `<id>.cond_mel.npy` - the prompt log-mel `extract_features` returned, float32 [1, 100, frames], written with
numpy.save - next to `<id>.meta.json` = {"id": <sanitised name>, "user_given_name": <name>}.  `infer(prompt_mel=...)` takes
the loaded array as is, so voices saved by the reference UI and by this package are interchangeable."""
from __future__ import annotations

import json
import os
import re

import numpy as np
import torch


def sanitize_filename(name) -> str:
    """The id the web UI derives from a user-given voice name: word characters, dots and hyphens; blanks -> '-'."""
    name = re.sub(r"[^\w\s.-]", "", str(name)).strip()
    return re.sub(r"[-\s]+", "-", name).replace("/", "_").replace("\\", "_")


def save_voice(directory: str, user_given_name: str, cond_mel) -> str:
    """Write `<id>.cond_mel.npy` + `<id>.meta.json`; returns the id."""
    vid = sanitize_filename(user_given_name)
    if not vid:
        raise ValueError("voice name is empty after sanitising")
    mel = cond_mel.detach().cpu().numpy() if isinstance(cond_mel, torch.Tensor) else np.asarray(cond_mel)
    if mel.ndim != 3 or mel.shape[0] != 1:
        raise ValueError(f"cond_mel must be [1, n_mels, frames], got {mel.shape}")
    os.makedirs(directory, exist_ok=True)
    np.save(os.path.join(directory, f"{vid}.cond_mel.npy"), mel)
    with open(os.path.join(directory, f"{vid}.meta.json"), "w", encoding="utf-8") as f:
        json.dump({"id": vid, "user_given_name": user_given_name}, f, ensure_ascii=False, indent=2)
    return vid


def load_voice(directory: str, voice_id: str, device=None) -> torch.Tensor:
    """`<id>.cond_mel.npy` -> prompt_mel tensor [1, n_mels, frames] (what webui.py:311-313 does)."""
    path = os.path.join(directory, f"{sanitize_filename(voice_id)}.cond_mel.npy")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Saved voice '{voice_id}' not found.")
    mel = torch.from_numpy(np.load(path))
    return mel.to(device) if device is not None else mel


def list_voices(directory: str):
    out = []
    for fn in sorted(os.listdir(directory)) if os.path.isdir(directory) else []:
        if fn.endswith(".meta.json"):
            with open(os.path.join(directory, fn), encoding="utf-8") as f:
                out.append(json.load(f))
    return out
