"""Model configuration for the IndexTTS hot path.

Mirrors the sections of the checkpoint's ``config.yaml`` that the reference reads
(``/root/reference/indextts/infer.py:42-69``: ``cfg.gpt``, ``cfg.bigvgan``, ``cfg.vqvae``,
``cfg.dataset.bpe_model``, ``cfg.gpt_checkpoint``, ``cfg.bigvgan_checkpoint``).  The reference loads the
file with OmegaConf; OmegaConf is only a loader, so this module reads the same YAML with ``yaml`` and
exposes attribute *and* item access (``cfg.gpt.model_dim`` / ``cfg["gpt"]["model_dim"]``) the way the
reference code uses it.  No dimension is hard-coded in the engine: everything flows from here.
"""
from __future__ import annotations

import copy
from typing import Any, Dict


class Node(dict):
    """dict with attribute access (the subset of OmegaConf behaviour the reference relies on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, d=None):  # noqa: D401  (same as dict.get, kept explicit for readability)
        return dict.get(self, k, d)


def to_node(x: Any) -> Any:
    if isinstance(x, dict):
        return Node({k: to_node(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return [to_node(v) for v in x]
    return x


# IndexTTS-1.5 values (public IndexTeam/IndexTTS-1.5 config.yaml).
_INDEXTTS_1_5: Dict[str, Any] = {
    "dataset": {"bpe_model": "bpe.model", "sample_rate": 24000,
                "mel": {"sample_rate": 24000, "n_fft": 1024, "hop_length": 256, "win_length": 1024,
                        "n_mels": 100, "mel_fmin": 0, "normalize": False}},
    "gpt": {
        "model_dim": 1280, "max_mel_tokens": 800, "max_text_tokens": 600, "heads": 20,
        "use_mel_codes_as_input": True, "mel_length_compression": 1024, "layers": 24,
        "number_text_tokens": 12000, "number_mel_codes": 8194, "start_mel_token": 8192,
        "stop_mel_token": 8193, "start_text_token": 0, "stop_text_token": 1,
        "train_solo_embeddings": False, "condition_type": "conformer_perceiver",
        "condition_module": {"output_size": 512, "linear_units": 2048, "attention_heads": 8,
                             "num_blocks": 6, "input_layer": "conv2d2", "perceiver_mult": 2},
    },
    "vqvae": {"channels": 100, "num_tokens": 8192, "hidden_dim": 512, "num_resnet_blocks": 3,
              "codebook_dim": 512, "num_layers": 2, "positional_dims": 1, "kernel_size": 3,
              "smooth_l1_loss": True, "use_transposed_convs": False},
    "bigvgan": {
        "adam_b1": 0.8, "adam_b2": 0.99, "lr_decay": 0.999998, "seed": 1234,
        "resblock": "1", "upsample_rates": [4, 4, 4, 4, 2, 2], "upsample_kernel_sizes": [8, 8, 4, 4, 4, 4],
        "upsample_initial_channel": 1536, "resblock_kernel_sizes": [3, 7, 11],
        "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "feat_upsample": False,
        "speaker_embedding_dim": 512, "cond_d_vector_in_each_upsampling_layer": True,
        "gpt_dim": 1280, "activation": "snakebeta", "snake_logscale": True,
        "use_cqtd_instead_of_mrd": True, "num_mels": 100, "n_fft": 1024, "hop_size": 256,
        "win_size": 1024, "sampling_rate": 24000, "fmin": 0, "fmax": None,
    },
    "gpt_checkpoint": "gpt.pth", "dvae_checkpoint": "dvae.pth", "bigvgan_checkpoint": "bigvgan_generator.pth",
    "version": 1.5,
}

# A micro configuration with the *real topology* (6 vocoder stages x 3 AMP kernels, conformer +
# perceiver conditioning, left-pad batching) at toy widths: used for golden fixtures and CPU-fast tests.
_MICRO: Dict[str, Any] = copy.deepcopy(_INDEXTTS_1_5)
_MICRO["gpt"].update({
    "model_dim": 128, "heads": 2, "layers": 2, "max_mel_tokens": 60, "max_text_tokens": 40,
    "number_text_tokens": 64, "number_mel_codes": 66, "start_mel_token": 64, "stop_mel_token": 65,
    "condition_module": {"output_size": 64, "linear_units": 128, "attention_heads": 2,
                         "num_blocks": 2, "input_layer": "conv2d2", "perceiver_mult": 2},
})
# ECAPA keeps the reference's hard-coded widths (BigVGAN.__init__ builds it with defaults, models.py:191).
_MICRO["bigvgan"].update({"upsample_initial_channel": 512, "gpt_dim": 128, "speaker_embedding_dim": 32})
_MICRO["vqvae"].update({"num_tokens": 66, "hidden_dim": 32, "codebook_dim": 32})
_MICRO["version"] = "micro"


def indextts_1_5() -> Node:
    return to_node(copy.deepcopy(_INDEXTTS_1_5))


def micro() -> Node:
    return to_node(copy.deepcopy(_MICRO))


def load_yaml(path: str) -> Node:
    """Read a reference ``config.yaml`` (what ``OmegaConf.load`` does at infer.py:42)."""
    import yaml

    with open(path, "r", encoding="utf-8") as f:
        raw = yaml.safe_load(f)
    return to_node(raw)


def dump_yaml(cfg: Dict[str, Any], path: str) -> None:
    import yaml

    def plain(x):
        if isinstance(x, dict):
            return {k: plain(v) for k, v in x.items()}
        if isinstance(x, list):
            return [plain(v) for v in x]
        return x

    with open(path, "w", encoding="utf-8") as f:
        yaml.safe_dump(plain(cfg), f, sort_keys=False)


# ---- derived quantities used by both the oracle and the engine ------------------------------------

def ecapa_dims(bv: Dict[str, Any]) -> Dict[str, Any]:
    """ECAPA-TDNN hyper-parameters.  The reference hard-codes its defaults
    (ECAPA_TDNN.py:429-449: channels [512,512,512,512,1536], kernels [5,3,3,3,1], dilations [1,2,3,4,1],
    attention 128, res2net scale 8, se 128); only the micro config overrides them."""
    return {
        "channels": list(bv.get("ecapa_channels", [512, 512, 512, 512, 1536])),
        "kernel_sizes": [5, 3, 3, 3, 1],
        "dilations": [1, 2, 3, 4, 1],
        "attention_channels": int(bv.get("ecapa_attention_channels", 128)),
        "res2net_scale": int(bv.get("ecapa_res2net_scale", 8)),
        "se_channels": int(bv.get("ecapa_se_channels", 128)),
        "lin_neurons": int(bv["speaker_embedding_dim"]),
        "input_size": int(bv["num_mels"]),
    }


def perceiver_inner(gpt: Dict[str, Any]) -> int:
    """GEGLU inner width: int(dim * mult * 2 / 3) (perceiver.py FeedForward)."""
    return int(gpt["model_dim"] * gpt["condition_module"]["perceiver_mult"] * 2 / 3)
