"""MI355X drop-in for the reference's ``audio_utilities.py``.

Same function names, positional arguments and return shapes as
/root/reference/audio_utilities.py; the numpy loops are replaced by the HIP
kernels behind include/gl_hip.h.  Extensions are keyword-only and default to the
reference's behaviour:

* numpy in -> numpy out (float64 / complex128 like the reference), torch in ->
  torch out (fp32 / complex64, left on the GPU);
* a leading batch dimension is accepted (``x`` (B, L), spectrograms (B, T, F));
* ``x_init`` / ``seed`` / ``rng`` choose the start signal of Griffin-Lim; by
  default it is drawn with ``np.random.randn`` from numpy's global legacy RNG,
  exactly as audio_utilities.py:176 does, so ``np.random.seed(s)`` reproduces
  the reference's initial phase.

There is no CPU fallback: without the HIP library or a GPU these functions raise.
"""
import wave

import numpy as np
import torch

from .plan import get_plan, _require_gpu

__all__ = [
    "hz_to_mel", "mel_to_hz", "fft_bin_to_hz", "hz_to_fft_bin", "stft_for_reconstruction",
    "istft_for_reconstruction", "get_signal", "reconstruct_signal_griffin_lim", "save_audio_to_file",
    "pcm16_from_float",
]


# -- small host helpers (reference audio_utilities.py:14-74; unused by the hot path) ------
def hz_to_mel(f_hz):
    """O'Shaughnessy (HTK) mel scale, audio_utilities.py:14-24."""
    return 2595 * np.log10(1.0 + f_hz / 700.0)


def mel_to_hz(m_mel):
    """Inverse of hz_to_mel, audio_utilities.py:27-37."""
    return 700 * (10 ** (m_mel / 2595) - 1.0)


def fft_bin_to_hz(n_bin, sample_rate_hz, fft_size):
    """audio_utilities.py:40-54 (note the reference's 2*fft_size denominator)."""
    return float(n_bin) * float(sample_rate_hz) / (2.0 * float(fft_size))


def hz_to_fft_bin(f_hz, sample_rate_hz, fft_size):
    """audio_utilities.py:57-74."""
    fft_size = float(fft_size)
    fft_bin = int(np.round(float(f_hz) * 2.0 * fft_size / float(sample_rate_hz)))
    if fft_bin >= fft_size:
        fft_bin = fft_size - 1
    return fft_bin


def _read_wav(in_file):
    """(sample rate, mono float samples) of a 16-bit / 32-bit PCM or float32 wav file."""
    import scipy.io.wavfile
    fs, y = scipy.io.wavfile.read(in_file)
    kind = y[0].dtype
    if kind == 'int16':
        y = y * (1.0 / 32768)
    elif kind == 'int32':
        y = y * (1.0 / 2147483648)
    elif kind == 'float32':
        pass
    elif kind == 'uint8':
        raise Exception('8-bit PCM is not supported.')
    else:
        raise Exception('Unknown format.')
    return fs, (y if y.ndim == 1 else y.mean(axis=1))


def get_signal(in_file, expected_fs=44100):
    """Load a wav file as mono float, audio_utilities.py:117-151."""
    fs, y = _read_wav(in_file)
    if fs != expected_fs:
        raise Exception('Invalid sample rate.')
    return y


# -- tensor plumbing ------------------------------------------------------------------------
def _to_device(a, device, dtype):
    """numpy / torch -> contiguous torch tensor of `dtype` on `device`; returns (tensor, was_numpy)."""
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous(), False
    arr = np.asarray(a)
    if np.iscomplexobj(arr):
        arr = arr.astype(np.complex64)
    else:
        arr = arr.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device=device, dtype=dtype), True


def _pick_device(device, *tensors):
    if device is None:
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                return _require_gpu(t.device)
    return _require_gpu(device)


# -- a-4 ---------------------------------------------------------------------------------------
def stft_for_reconstruction(x, fft_size, hopsamp, *, device=None):
    """STFT with symmetric Hann, no padding (audio_utilities.py:77-92).

    x: (L,) or (B, L).  Returns (T, F) [or (B, T, F)], rows = time slices,
    T = len(range(0, L - fft_size, hopsamp))."""
    fft_size, hopsamp = int(fft_size), int(hopsamp)
    dev = _pick_device(device, x)
    xt, was_np = _to_device(x, dev, torch.float32)
    single = xt.dim() == 1
    if single:
        xt = xt[None, :]
    if xt.dim() != 2:
        raise ValueError("x must be 1-D or 2-D (batch, samples)")
    B, L = xt.shape
    T = len(range(0, L - fft_size, hopsamp))
    plan = get_plan(fft_size, hopsamp, max(T, 1), B, dev)
    S = plan.stft(xt, frames=T)
    if single:
        S = S[0]
    if was_np:
        return S.cpu().numpy().astype(np.complex128)
    return S


# -- a-5 ---------------------------------------------------------------------------------------
def istft_for_reconstruction(X, fft_size, hopsamp, *, normalize=False, device=None):
    """Inverse STFT by windowed overlap-add without window-sum normalisation
    (audio_utilities.py:95-114).  X: (T, F) or (B, T, F) complex.
    Returns (T*hopsamp + fft_size,) [or (B, ...)].  normalize=True (not the reference) divides by
    the overlap-added squared window."""
    fft_size, hopsamp = int(fft_size), int(hopsamp)
    dev = _pick_device(device, X)
    Xt, was_np = _to_device(X, dev, torch.complex64)
    single = Xt.dim() == 2
    if single:
        Xt = Xt[None]
    if Xt.dim() != 3:
        raise ValueError("X must be (T, F) or (B, T, F)")
    B, T, F = Xt.shape
    if F != fft_size // 2 + 1:
        raise ValueError(f"X has {F} bins, fft_size {fft_size} needs {fft_size // 2 + 1}")
    if T == 0:
        y = torch.zeros((B, fft_size), dtype=torch.float32, device=dev)
    else:
        plan = get_plan(fft_size, hopsamp, T, B, dev)
        y = plan.istft(Xt, normalize=normalize)
    if single:
        y = y[0]
    if was_np:
        return y.cpu().numpy().astype(np.float64)
    return y


def _zero_past(x, lengths, hop, n_fft):
    """Samples past an utterance's own length (lengths[b]*hop + n_fft) are padding: exact zeros."""
    for b, t in enumerate(lengths):
        x[b, int(t) * hop + n_fft:].zero_()


# -- a-3 / a-6 ---------------------------------------------------------------------------------
def _start_signal(B, L, x_init, seed, rng, plan, dev):
    if x_init is not None:
        x0, _ = _to_device(x_init, dev, torch.float32)
        if x0.dim() == 1:
            x0 = x0[None, :]
        if tuple(x0.shape) != (B, L):
            raise ValueError(f"x_init must have shape ({B}, {L}) (or ({L},) for one utterance)")
        return x0.clone() if isinstance(x_init, torch.Tensor) else x0
    if rng == "device":
        return plan.randn((B, L), 0 if seed is None else seed)
    if rng != "numpy":
        raise ValueError("rng must be 'numpy' or 'device'")
    # audio_utilities.py:176 -- np.random.randn(len_samples) from the global legacy RNG,
    # one draw per utterance in batch order
    draw = np.random.randn if seed is None else np.random.RandomState(seed).randn
    host = np.stack([draw(L) for _ in range(B)]).astype(np.float32)
    return torch.from_numpy(host).to(dev)


def reconstruct_signal_griffin_lim(magnitude_spectrogram, fft_size, hopsamp, iterations, *,
                                   x_init=None, seed=None, rng="numpy", device=None, normalize=False,
                                   momentum=0.0, lengths=None, path=None, run_frames=None):
    """Griffin-Lim reconstruction (audio_utilities.py:154-188).

    magnitude_spectrogram: (T, F) or (B, T, F), rows = time slices.  The whole
    loop (STFT -> keep phase, impose magnitude -> ISTFT) runs on the GPU on the
    current HIP stream.  Returns (T*hopsamp + fft_size,) [or (B, ...)].
    normalize / momentum (both off = the reference): window-sum normalised inverse STFT in every
    iteration; fast Griffin-Lim extrapolation x_n + momentum (x_n - x_{n-1}).
    lengths: frames per utterance of a ragged batch (B,); rows beyond them are ignored and utterance b
    comes out as it would alone with lengths[b] frames, followed by zeros (see gl_mask_frames).
    path / run_frames: for this call only -- 1 forces the generic any-N kernels; run_frames pins the fused
    kernels' frames-per-wave-run, which makes results bit-identical across batch compositions (shards of a
    batch vs the whole batch, chunks of a long utterance vs one piece)."""
    fft_size, hopsamp, iterations = int(fft_size), int(hopsamp), int(iterations)
    dev = _pick_device(device, magnitude_spectrogram, x_init)
    mag, was_np = _to_device(magnitude_spectrogram, dev, torch.float32)
    single = mag.dim() == 2
    if single:
        mag = mag[None]
    if mag.dim() != 3:
        raise ValueError("magnitude_spectrogram must be (T, F) or (B, T, F)")
    B, T, F = mag.shape
    if F != fft_size // 2 + 1:
        raise ValueError(f"spectrogram has {F} bins, fft_size {fft_size} needs {fft_size // 2 + 1}")
    if T < 1:
        raise ValueError("need at least one frame")
    L = T * hopsamp + fft_size
    plan = get_plan(fft_size, hopsamp, T, B, dev)
    if lengths is not None and normalize:
        raise ValueError("normalize=True needs the window sum of each utterance's own length: run ragged "
                         "batches without it or one length at a time")
    if lengths is not None:
        if mag.data_ptr() == getattr(magnitude_spectrogram, "data_ptr", lambda: 0)():
            mag = mag.clone()            # never modify the caller's tensor
        plan.mask_frames(mag, lengths)
    x0 = _start_signal(B, L, x_init, seed, rng, plan, dev)
    x = plan.run(mag, x0, iterations, out=x0, normalize=normalize, momentum=momentum, path=path,
                 run_frames=run_frames)
    if lengths is not None:
        _zero_past(x, lengths, hopsamp, fft_size)
    if single:
        x = x[0]
    if was_np:
        return x.cpu().numpy().astype(np.float64)
    return x


# -- a-7 ---------------------------------------------------------------------------------------
def pcm16_from_float(x):
    """Sample conversion of save_audio_to_file (audio_utilities.py:202-210):
    assert max|x| <= 1, rescale to full scale, x*32767, round half to even."""
    x = np.array(x, dtype=np.float64, copy=True)
    x_max = np.max(abs(x))
    assert x_max <= 1.0, 'Input audio value is out of range. Should be in the range [-1.0, 1.0].'
    if x_max == 0 or not np.isfinite(x_max):
        # the reference's x /= 0 gives NaN and int(round(nan)) raises (audio_utilities.py:204-209)
        raise ValueError("cannot convert float NaN to integer")
    x /= x_max
    x *= 32767.0
    return np.rint(x).astype('<i2')


def pcm16_encode_device(x):
    """Device form of the sample conversion (N2): x is a CUDA fp32 tensor (L,) or (B, L); returns
    a little-endian int16 numpy array of the same shape.  Only the 2-byte samples cross PCIe.
    Raises like the reference: AssertionError when max|x| > 1 (audio_utilities.py:203), ValueError
    when max|x| == 0 (the reference's x /= 0 gives NaN and int(round(nan)) raises)."""
    from .plan import get_plan
    one = (x.dim() == 1)
    xb = (x[None] if one else x).contiguous().to(torch.float32)
    B, L = xb.shape
    plan = get_plan(64, 16, 1, B, xb.device)      # geometry is irrelevant to the encode kernels
    pcm, peaks = plan.pcm16_encode(xb)
    peaks = peaks.cpu().numpy()
    assert float(peaks.max()) <= 1.0, 'Input audio value is out of range. Should be in the range [-1.0, 1.0].'
    if float(peaks.min()) == 0.0:
        raise ValueError("cannot convert float NaN to integer")
    out = pcm.cpu().numpy().astype('<i2', copy=False)
    return out[0] if one else out


def save_audio_to_file(x, sample_rate, outfile='out.wav'):
    """Write a mono 16-bit PCM WAV (audio_utilities.py:191-215).  A CUDA tensor is encoded on the
    device (gl_pcm16_encode) and only the int16 samples are copied back; a numpy array / CPU tensor
    is encoded on the host (the per-sample Python loop of the reference, vectorised).  Same bytes
    either way.  Unlike the reference, `x` is not modified in place."""
    if isinstance(x, torch.Tensor) and x.is_cuda:
        print(float(x.max()))
        print(float(x.min()))
        data = pcm16_encode_device(x)
    else:
        if isinstance(x, torch.Tensor):
            x = x.detach().cpu().numpy()
        print(np.max(x))
        print(np.min(x))
        data = pcm16_from_float(x)
    f = wave.open(outfile, 'w')
    f.setparams((1, 2, int(sample_rate), 0, "NONE", "Uncompressed"))
    f.writeframes(data.tobytes())
    f.close()
