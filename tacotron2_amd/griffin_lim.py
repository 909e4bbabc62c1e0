"""MI355X drop-in for the reference's ``griffin_lim.py``.

``save_mel_to_wav(mel_spectrogram, filename='out')`` keeps the reference
signature and behaviour (griffin_lim.py:17-35): mel -> linear magnitude with the
TRANSPOSE of the mel filterbank (not a pseudo-inverse), Griffin-Lim with the
``params`` geometry, peak-normalise only if max|x| > 1, write
``<audio_out_path>/<filename>.wav``, return the (F, T) linear magnitude.
The contraction runs on MFMA (gl_mel_to_mag) and the loop on the fused HIP
kernels (gl_run); the 16-bit sample conversion runs on the device as well (gl_pcm16_encode), the
host only writes the WAV container.
"""
import os
from os.path import join

import numpy as np
import torch

from . import audio_utilities, filters
from .plan import get_plan, _require_gpu

try:  # the caller's own configuration module wins (infer.py does `import params`)
    import params as _params
    for _name in ("sampling_rate", "n_fft", "n_mel_channels", "f_min", "f_max", "frame_hop",
                  "griffin_lim_iters", "audio_out_path"):
        getattr(_params, _name)
except (ImportError, AttributeError):
    from . import params as _params
params = _params

# griffin_lim.py:13-14 -- module-level filterbank, shape (n_mels, 1 + n_fft//2), float32
filterbank = filters.mel(sr=params.sampling_rate, n_fft=params.n_fft, n_mels=params.n_mel_channels,
                         fmin=params.f_min, fmax=params.f_max)

_fb_cache = {}


def _device_filterbank(fb, dev):
    key = (id(fb), dev.index)
    hit = _fb_cache.get(key)
    if hit is not None and hit[0] is fb:
        return hit[1]
    t = torch.from_numpy(np.ascontiguousarray(fb, dtype=np.float32)).to(dev)
    _fb_cache[key] = (fb, t)
    return t


def mel_to_audio(mel_spectrogram, *, basis=None, n_fft=None, hop=None, iterations=None, x_init=None, seed=None,
                 rng="numpy", normalize="reference", device=None, window_sum_normalize=False, momentum=0.0,
                 lengths=None, run_frames=None, return_mag=True):
    """Batched core of save_mel_to_wav, everything on the GPU.

    mel_spectrogram: (n_mels, T) or (B, n_mels, T), numpy or torch (linear-amplitude
    mel, i.e. after exp, infer.py:27).  basis: (F, n_mels) inversion basis; default
    ``filterbank.T`` like griffin_lim.py:18; ``"pinv"`` = pseudo-inverse of the filterbank with the
    magnitudes clamped at 0 (north-star variant, not the reference).  window_sum_normalize /
    momentum / lengths (ragged batch: frames per utterance): see
    audio_utilities.reconstruct_signal_griffin_lim.  normalize: 'reference' (divide by the
    peak only when it exceeds 1, griffin_lim.py:27-29), 'peak' (always) or None.  run_frames: pin the
    frames-per-wave-run for this call (shards of one batch then reproduce the whole batch bit for bit).
    return_mag=False: the caller does not want the linear magnitude save_mel_to_wav returns (griffin_lim.py:35);
    where the plan has it the inversion and the loop are then ONE library call (gl_mel_run) that never
    materialises the (B, T, F) array -- same samples, bit for bit.
    Returns (waveforms (B, L) fp32 torch on device, magnitude (B, T, F) fp32 torch or None)."""
    n_fft = int(params.n_fft if n_fft is None else n_fft)
    hop = int(params.frame_hop if hop is None else hop)
    iterations = int(params.griffin_lim_iters if iterations is None else iterations)
    dev = audio_utilities._pick_device(device, mel_spectrogram)
    mel, _ = audio_utilities._to_device(mel_spectrogram, dev, torch.float32)
    if mel.dim() == 2:
        mel = mel[None]
    if mel.dim() != 3:
        raise ValueError("mel_spectrogram must be (n_mels, T) or (B, n_mels, T)")
    B, n_mels, T = mel.shape
    clamp = isinstance(basis, str)
    if clamp:
        if basis != "pinv":
            raise ValueError("basis must be an (F, n_mels) array, None or 'pinv'")
        fb = filterbank if (n_fft == params.n_fft and n_mels == params.n_mel_channels) else filters.mel(
            sr=params.sampling_rate, n_fft=n_fft, n_mels=n_mels, fmin=params.f_min, fmax=params.f_max)
        basis = filters.pinv_basis(fb)
    if basis is None:
        if n_fft != params.n_fft:
            fb = filters.mel(sr=params.sampling_rate, n_fft=n_fft, n_mels=n_mels, fmin=params.f_min,
                             fmax=params.f_max)
            fb_dev = torch.from_numpy(fb).to(dev)
        else:
            fb_dev = _device_filterbank(filterbank, dev)
    else:
        b, _ = audio_utilities._to_device(basis, dev, torch.float32)
        fb_dev = b.t().contiguous()          # kernel wants (n_mels, F)
    if tuple(fb_dev.shape) != (n_mels, n_fft // 2 + 1):
        raise ValueError(f"basis/filterbank shape {tuple(fb_dev.shape)} does not match "
                         f"({n_mels}, {n_fft // 2 + 1})")
    plan = get_plan(n_fft, hop, T, B, dev)
    if lengths is not None and window_sum_normalize:
        raise ValueError("window_sum_normalize needs each utterance's own window sum: not with lengths=")
    L = plan.signal_length(T)
    x0 = audio_utilities._start_signal(B, L, x_init, seed, rng, plan, dev)
    if not return_mag and plan.mel_run_supported():
        mag = None
        x = plan.mel_run(fb_dev, mel.contiguous(), x0, iterations, out=x0, clamp=clamp, lengths=lengths,
                         normalize=window_sum_normalize, momentum=momentum, run_frames=run_frames)   # :18-24
    else:
        mag = plan.mel_to_mag(fb_dev, mel, clamp=clamp)          # griffin_lim.py:18-19
        if lengths is not None:
            plan.mask_frames(mag, lengths)
        x = plan.run(mag, x0, iterations, out=x0, normalize=window_sum_normalize, momentum=momentum,
                     run_frames=run_frames)                                                          # :23-24
    if lengths is not None:
        audio_utilities._zero_past(x, lengths, hop, n_fft)
    if normalize == "reference":
        plan.peak_normalise(x, mode=0)                       # griffin_lim.py:27-29
    elif normalize == "peak":
        plan.peak_normalise(x, mode=1)
    elif normalize is not None:
        raise ValueError("normalize must be 'reference', 'peak' or None")
    return x, (mag if return_mag else None)


def logmel_to_audio(y_pred_post, *, compression_factor=1.0, basis=None, n_fft=None, hop=None, iterations=None,
                    x_init=None, seed=None, rng="numpy", normalize="reference", device=None, lengths=None,
                    return_mag=True):
    """infer.py:24-29 without the host round trip: the decoder output goes in as it is.

    y_pred_post: (B, T, n_mels) or (T, n_mels) natural-log mel (Tacotron2.inference's second
    result), numpy or torch.  dynamic_range_decompression (data.py:84-85), the transpose of
    infer.py:28 and the filterbank contraction run in ONE kernel (gl_logmel_to_mag); then Griffin-Lim
    and the normalisation of save_mel_to_wav.  Unlike the reference (batch forced to 1,
    model.py:530) any batch size is accepted.  return_mag=False: see mel_to_audio.
    Returns (waveforms (B, L), magnitude (B, T, F) or None)."""
    n_fft = int(params.n_fft if n_fft is None else n_fft)
    hop = int(params.frame_hop if hop is None else hop)
    iterations = int(params.griffin_lim_iters if iterations is None else iterations)
    dev = audio_utilities._pick_device(device, y_pred_post)
    lm, _ = audio_utilities._to_device(y_pred_post, dev, torch.float32)
    if lm.dim() == 2:
        lm = lm[None]
    if lm.dim() != 3:
        raise ValueError("y_pred_post must be (T, n_mels) or (B, T, n_mels)")
    lm = lm.contiguous()
    B, T, n_mels = lm.shape
    if basis is None:
        if n_fft != params.n_fft or n_mels != params.n_mel_channels:
            fb_dev = torch.from_numpy(filters.mel(sr=params.sampling_rate, n_fft=n_fft, n_mels=n_mels,
                                                  fmin=params.f_min, fmax=params.f_max)).to(dev)
        else:
            fb_dev = _device_filterbank(filterbank, dev)
    else:
        b, _ = audio_utilities._to_device(basis, dev, torch.float32)
        fb_dev = b.t().contiguous()
    if tuple(fb_dev.shape) != (n_mels, n_fft // 2 + 1):
        raise ValueError(f"basis/filterbank shape {tuple(fb_dev.shape)} does not match "
                         f"({n_mels}, {n_fft // 2 + 1})")
    plan = get_plan(n_fft, hop, T, B, dev)
    L = plan.signal_length(T)
    x0 = audio_utilities._start_signal(B, L, x_init, seed, rng, plan, dev)
    if not return_mag and plan.mel_run_supported():
        mag = None
        x = plan.mel_run(fb_dev, lm, x0, iterations, out=x0, logmel=True, compression_factor=compression_factor,
                         lengths=lengths)
    else:
        mag = plan.logmel_to_mag(fb_dev, lm, compression_factor)
        if lengths is not None:             # ragged batch: decoder frames past an utterance's end are ignored
            plan.mask_frames(mag, lengths)
        x = plan.run(mag, x0, iterations, out=x0)
    if lengths is not None:
        audio_utilities._zero_past(x, lengths, hop, n_fft)
    if normalize == "reference":
        plan.peak_normalise(x, mode=0)
    elif normalize == "peak":
        plan.peak_normalise(x, mode=1)
    elif normalize is not None:
        raise ValueError("normalize must be 'reference', 'peak' or None")
    return x, (mag if return_mag else None)


def infer_to_wav(y_pred_post, filenames=None, **kw):
    """Batched counterpart of infer.py:27-29: log-mel batch -> one WAV per utterance under
    params.audio_out_path ("Inference.wav" for a single utterance, like the reference).  The PCM
    encode runs on the device too (gl_pcm16_encode).  Returns the list of paths."""
    x, _ = logmel_to_audio(y_pred_post, return_mag=False, **kw)
    B = x.shape[0]
    if filenames is None:
        filenames = ["Inference"] if B == 1 else ["Inference_%d" % i for i in range(B)]
    if len(filenames) != B:
        raise ValueError("one filename per utterance")
    out_dir = params.audio_out_path
    os.makedirs(out_dir, exist_ok=True)
    pcm = audio_utilities.pcm16_encode_device(x)
    paths = []
    import wave
    for i, name in enumerate(filenames):
        path = join(out_dir, name + '.wav')
        f = wave.open(path, 'w')
        f.setparams((1, 2, int(params.sampling_rate), 0, "NONE", "Uncompressed"))
        f.writeframes(pcm[i].tobytes())
        f.close()
        paths.append(path)
    return paths


def save_mel_to_wav(mel_spectrogram, filename='out', *, basis=None, iterations=None, x_init=None, seed=None,
                    rng="numpy", device=None, write_wav=True):
    """Reference entry point (griffin_lim.py:17-35); see the module docstring.

    mel_spectrogram: (n_mels, T) array (any strides; infer.py:28 passes a transposed view)."""
    if np.ndim(mel_spectrogram) != 2:
        raise ValueError("save_mel_to_wav takes one (n_mels, T) spectrogram; use mel_to_audio for batches")
    x, mag = mel_to_audio(mel_spectrogram, basis=basis, iterations=iterations, x_init=x_init, seed=seed, rng=rng,
                          normalize="reference", device=device)
    if write_wav:
        out_dir = params.audio_out_path
        os.makedirs(out_dir, exist_ok=True)
        audio_utilities.save_audio_to_file(x[0], params.sampling_rate, outfile=join(out_dir, filename + '.wav'))
    # griffin_lim.py:35 returns stft_modified.T == the (F, T) linear magnitude
    lin = mag[0].t()
    if isinstance(mel_spectrogram, torch.Tensor):
        return lin
    return lin.cpu().numpy()
