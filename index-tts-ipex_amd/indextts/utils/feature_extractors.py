"""Log-mel front end with the reference's parameters
(/root/reference/indextts/utils/feature_extractors.py:24-50: torchaudio MelSpectrogram(24 kHz, n_fft 1024, hop 256,
100 mels, center=True, power=1) followed by safe_log(clip 1e-7)).  torchaudio is not a dependency here: the STFT is
torch.stft and the filterbank is the HTK-scale, un-normalised triangular bank torchaudio builds by default.
`resample` restates torchaudio.transforms.Resample's published algorithm (Hann-windowed sinc polyphase bank).
Runs once per prompt on the host.  PARITY UNPINNED: torchaudio is absent offline and the reference holds no
mel values, so this file is checked analytically only (tests/test_host_logic.py, tests/test_frontend_cpu.py: filterbank / STFT
identities, resampler tone and DC tests) - not against torchaudio outputs."""
from __future__ import annotations

import math

import torch


def _hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)  # [n_freqs, n_mels]


class MelSpectrogramFeatures(torch.nn.Module):
    def __init__(self, sample_rate=24000, n_fft=1024, hop_length=256, n_mels=100, padding="center"):
        super().__init__()
        if padding not in ("center", "same"):
            raise ValueError("Padding must be 'center' or 'same'.")
        self.padding, self.n_fft, self.hop = padding, n_fft, hop_length
        self.register_buffer("window", torch.hann_window(n_fft, periodic=True))
        self.register_buffer("fb", mel_filterbank(n_fft // 2 + 1, 0.0, sample_rate / 2.0, n_mels, sample_rate))

    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        """audio [B, samples] -> log-mel [B, n_mels, frames]."""
        if self.padding == "same":
            pad = self.n_fft - self.hop
            audio = torch.nn.functional.pad(audio, (pad // 2, pad // 2), mode="reflect")
        spec = torch.stft(audio, self.n_fft, self.hop, self.n_fft, self.window, center=self.padding == "center",
                          pad_mode="reflect", return_complex=True).abs()
        mel = torch.matmul(spec.transpose(1, 2), self.fb).transpose(1, 2)
        return torch.log(torch.clip(mel, min=1e-7))


def load_wav_mono(path: str):
    """Minimal wav reader (PCM16/PCM32/float32) -> (mono float tensor [1, n], sample_rate)."""
    from scipy.io import wavfile

    sr, data = wavfile.read(path)
    x = torch.from_numpy(data.copy())
    if x.dtype == torch.int16:
        x = x.float() / 32768.0
    elif x.dtype == torch.int32:
        x = x.float() / 2147483648.0
    else:
        x = x.float()
    if x.ndim == 2:
        x = x.mean(dim=1)
    return x.unsqueeze(0), int(sr)


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """The polyphase windowed-sinc bank of torchaudio.transforms.Resample (defaults: "sinc_interp_hann",
    lowpass_filter_width 6, rolloff 0.99), which infer.py:88 applies to the prompt: for each of the new_freq output phases
    a Hann-windowed sinc low-pass at rolloff * min(orig, new) / 2, evaluated in float64 and stored as float32.
    Returns (kernels [new, 1, 2 * width + orig], width) for rates already divided by their gcd."""
    base = min(orig_freq, new_freq) * rolloff
    width = math.ceil(lowpass_filter_width * orig_freq / base)
    idx = torch.arange(-width, width + orig_freq, dtype=torch.float64)[None, None] / orig_freq
    t = torch.arange(0, -new_freq, -1, dtype=torch.float64)[:, None, None] / new_freq + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig_freq)
    return kernels.to(torch.float32), width


def resample(x: torch.Tensor, sr: int, new_sr: int) -> torch.Tensor:
    """torchaudio.transforms.Resample(sr, new_sr)(x) for x [channels, n] (infer.py:88): zero-pad by the filter half-width,
    strided convolution with the polyphase bank, interleave the phases, cut to ceil(new * n / orig) samples."""
    if sr == new_sr:
        return x
    g = math.gcd(int(sr), int(new_sr))
    orig, new = int(sr) // g, int(new_sr) // g
    kernels, width = sinc_resample_kernel(orig, new)
    shape = x.shape
    w = x.reshape(-1, shape[-1]).float()
    n = w.shape[-1]
    w = torch.nn.functional.pad(w, (width, width + orig))
    y = torch.nn.functional.conv1d(w[:, None], kernels, stride=orig)
    y = y.transpose(1, 2).reshape(w.shape[0], -1)
    target = int(math.ceil(new * n / orig))
    return y[..., :target].reshape(*shape[:-1], target)
