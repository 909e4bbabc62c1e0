"""Forward analysis path of the reference's data pipeline on the GPU.

``LabelledMelDataset.audiopath_to_mel`` (data.py:100-116) loads a wav with librosa, takes
``librosa.stft(y, n_fft=params.n_fft)`` (so hop = n_fft // 4 = 256, NOT params.frame_hop = 275;
periodic Hann; centred, edges padded), the magnitude, the mel filterbank product and
``dynamic_range_compression`` (data.py:80-81).  Here the same chain runs in two HIP kernels
(gl_wav_to_logmel): LDS FFT with the padding folded into the frame gather, then the filterbank
contraction on MFMA with clip + log in its epilogue.  librosa is not installed and not pinned by the
reference, so this follows its documented algorithm (parity unpinned); ``pad_mode`` selects the
edge rule of librosa < 0.10 ('reflect', the reference's era) or >= 0.10 ('constant').
"""
import numpy as np
import torch

from . import audio_utilities, griffin_lim
from .plan import get_plan


def dynamic_range_compression(mel, compression_factor=1):
    """data.py:80-81 (host helper, numpy)."""
    return np.log(np.clip(mel, a_min=0.01, a_max=None) * compression_factor)


def audio_to_mel(y, *, n_fft=None, hop=None, filterbank=None, pad_mode="reflect", compression_factor=1.0,
                 return_magnitude=False, device=None):
    """Waveform(s) -> log-mel like audiopath_to_mel (data.py:106-116).

    y: (len,) or (B, len) in [-1, 1], numpy or torch.  Returns (n_mels, T) [or (B, n_mels, T)],
    T = 1 + len // hop, numpy float32 for numpy input (librosa's stft is complex64) or a device
    tensor; with return_magnitude also |stft| as (T, F) rows = frames."""
    params = griffin_lim.params
    n_fft = int(params.n_fft if n_fft is None else n_fft)
    hop = int(n_fft // 4 if hop is None else hop)          # librosa.stft default hop_length
    dev = audio_utilities._pick_device(device, y)
    yt, was_np = audio_utilities._to_device(y, dev, torch.float32)
    single = yt.dim() == 1
    if single:
        yt = yt[None]
    if yt.dim() != 2:
        raise ValueError("y must be (len,) or (B, len)")
    # data.py:107-108
    assert float(yt.min()) >= -1
    assert float(yt.max()) <= 1
    fb = griffin_lim.filterbank if filterbank is None else filterbank
    if filterbank is None and n_fft != params.n_fft:
        from . import filters
        fb = filters.mel(sr=params.sampling_rate, n_fft=n_fft, n_mels=params.n_mel_channels, fmin=params.f_min,
                         fmax=params.f_max)
    fb_dev, _ = audio_utilities._to_device(fb, dev, torch.float32)
    B, n = yt.shape
    plan = get_plan(n_fft, hop, 1 + n // hop, B, dev)
    logmel, mag = plan.wav_to_logmel(yt, fb_dev, pad_mode=pad_mode, compression_factor=compression_factor)
    if single:
        logmel, mag = logmel[0], mag[0]
    if was_np:
        logmel, mag = logmel.cpu().numpy(), mag.cpu().numpy()
    return (logmel, mag) if return_magnitude else logmel


def resample(y, orig_sr, target_sr, device=None):
    """librosa.resample(y, orig_sr, target_sr) with its default res_type of the reference's era ('kaiser_best' =
    resampy's band-limited sinc interpolation) on the GPU (gl_resample).  y: (n,) or (B, n), numpy or torch;
    returns ceil(n * target_sr / orig_sr) samples per row, float32, of the same kind as y.  Parity unpinned
    (neither librosa nor resampy is available): see gl_hip.h."""
    dev = audio_utilities._pick_device(device, y)
    yt, was_np = audio_utilities._to_device(y, dev, torch.float32)
    single = yt.dim() == 1
    if single:
        yt = yt[None]
    if yt.dim() != 2:
        raise ValueError("y must be (n,) or (B, n)")
    if int(orig_sr) == int(target_sr):
        out = yt.clone()
    else:
        params = griffin_lim.params
        plan = get_plan(params.n_fft, params.n_fft // 4, 1, 1, dev)         # any plan carries the filter table
        out = plan.resample(yt, orig_sr, target_sr)
    if single:
        out = out[0]
    return out.cpu().numpy() if was_np else out


def load(path, sr=22050, device=None):
    """librosa.load(path, sr=22050) for wav files (data.py:101): mono float32 at `sr`, resampled when the file's
    rate differs; sr=None keeps the file's rate.  Returns (y, sr) like librosa."""
    fs, y = audio_utilities._read_wav(str(path))
    y = np.asarray(y, dtype=np.float32)
    if sr is None or int(sr) == int(fs):
        return y, int(fs)
    return resample(y, fs, sr, device=device), int(sr)


def audiopath_to_mel(path, **kw):
    """data.py:100-116 for a 16-bit / 32-bit PCM or float wav: loaded, resampled to 22 050 Hz like librosa.load's
    default when the file has another rate (the reference then asserts params.sampling_rate, data.py:104)."""
    y, sr = load(path, sr=22050, device=kw.get("device"))
    assert sr == griffin_lim.params.sampling_rate                                                    # data.py:104
    return audio_to_mel(y, **kw)
