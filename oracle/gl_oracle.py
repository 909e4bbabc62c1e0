"""CPU oracle for the Griffin-Lim vocoder hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the reference algorithm
(AidenMoosa/Tacotron2: ``audio_utilities.py`` / ``griffin_lim.py``).  It is the
*checker* for the HIP product path; nothing under ``tacotron2_amd/`` or
``dropin/`` imports it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.

Parity status
-------------
* ``stft_loop / istft_loop / griffin_lim_loop`` (a-3, a-4, a-5, a-6) are PINNED:
  ``oracle/make_golden.py`` imports the real reference
  (``/root/reference/audio_utilities.py``) in the build container, checks this
  restatement against it to <= 1e-12 and writes the golden vectors under
  ``tests/golden/``.
* ``pcm16_from_float`` (a-7) is pinned by hand-built known answers: the
  reference's ``save_audio_to_file`` dies at ``array.tostring()`` on
  Python >= 3.9 (audio_utilities.py:214) so it cannot be run end to end.
* ``mel_filterbank_slaney`` (a-1) is PARITY UNPINNED: librosa is not installed,
  not vendored and not version-pinned by the reference; this follows librosa's
  documented algorithm (htk=False, norm='slaney').

Every function cites the reference file:line it follows.
"""
import numpy as np


# --------------------------------------------------------------------------
# a-4  stft_for_reconstruction            audio_utilities.py:77-92
# --------------------------------------------------------------------------
def stft_loop(x, fft_size, hopsamp):
    """Loop form, float64.  One rfft per frame, symmetric Hann (np.hanning,
    audio_utilities.py:88), no padding / centring; frame i covers
    x[i : i+fft_size] for i in range(0, len(x)-fft_size, hopsamp)
    (audio_utilities.py:91-92)."""
    fft_size = int(fft_size)
    hopsamp = int(hopsamp)
    w = np.hanning(fft_size)
    rows = []
    for start in range(0, len(x) - fft_size, hopsamp):
        rows.append(np.fft.rfft(w * x[start:start + fft_size]))
    return np.array(rows)


# --------------------------------------------------------------------------
# a-5  istft_for_reconstruction           audio_utilities.py:95-114
# --------------------------------------------------------------------------
def istft_loop(X, fft_size, hopsamp):
    """Loop form, float64.  x = zeros(T*hop + N); x[i:i+N] += w*real(irfft(X[n]))
    in frame order, NO window-sum normalisation (audio_utilities.py:108-114)."""
    fft_size = int(fft_size)
    hopsamp = int(hopsamp)
    w = np.hanning(fft_size)
    n_frames = X.shape[0]
    total = int(n_frames * hopsamp + fft_size)
    x = np.zeros(total)
    for n, start in enumerate(range(0, len(x) - fft_size, hopsamp)):
        x[start:start + fft_size] += w * np.real(np.fft.irfft(X[n]))
    return x


# --------------------------------------------------------------------------
# a-3 / a-6  reconstruct_signal_griffin_lim   audio_utilities.py:154-188
# --------------------------------------------------------------------------
def griffin_lim_loop(magnitude_spectrogram, fft_size, hopsamp, iterations, x0=None):
    """Line-faithful loop form (float64) including the dead per-iteration RMSE
    computed with the Python builtin ``sum`` (audio_utilities.py:186) -- this is
    the form timed as the CPU baseline, because it is what the reference does.

    x0=None draws the start signal from numpy's global legacy RNG exactly like
    audio_utilities.py:176; passing x0 replaces only that draw."""
    n_frames = magnitude_spectrogram.shape[0]
    total = int(n_frames * hopsamp + fft_size)
    if x0 is None:
        x = np.random.randn(total)
    else:
        x = np.asarray(x0, dtype=np.float64)
        assert x.shape == (total,)
    left = iterations
    while left > 0:
        left -= 1
        S = stft_loop(x, fft_size, hopsamp)
        theta = np.angle(S)
        proposal = magnitude_spectrogram * np.exp(1.0j * theta)
        prev = x
        x = istft_loop(proposal, fft_size, hopsamp)
        _rmse = np.sqrt(sum((x - prev) ** 2) / x.size)  # dead value, kept for timing fidelity
    return x


# --------------------------------------------------------------------------
# Vectorised forms (same maths, used where the loop form would be too slow
# for a unit test).  Checked against the loop forms in tests/.
# --------------------------------------------------------------------------
def frame_starts(total_len, fft_size, hopsamp):
    return np.arange(0, total_len - int(fft_size), int(hopsamp))


def stft_vec(x, fft_size, hopsamp, dtype=np.float64):
    fft_size = int(fft_size)
    x = np.asarray(x, dtype=dtype)
    w = np.hanning(fft_size).astype(dtype)
    starts = frame_starts(len(x), fft_size, hopsamp)
    idx = starts[:, None] + np.arange(fft_size)[None, :]
    return np.fft.rfft(w[None, :] * x[idx], axis=1)


def istft_vec(X, fft_size, hopsamp, dtype=np.float64):
    fft_size = int(fft_size)
    hopsamp = int(hopsamp)
    w = np.hanning(fft_size).astype(dtype)
    n_frames = X.shape[0]
    total = n_frames * hopsamp + fft_size
    frames = (w[None, :] * np.fft.irfft(X, n=fft_size, axis=1)).astype(dtype)
    x = np.zeros(total, dtype=dtype)
    # frame order is preserved so the floating-point summation order matches
    # the reference's `+=` loop
    for n in range(n_frames):
        x[n * hopsamp:n * hopsamp + fft_size] += frames[n]
    return x


def project(S, mag):
    """audio_utilities.py:181-183: mag * exp(1j*angle(S)); angle(0) == 0."""
    return mag * np.exp(1.0j * np.angle(S))


def griffin_lim_vec(mag, fft_size, hopsamp, iterations, x0):
    x = np.asarray(x0, dtype=np.float64)
    for _ in range(int(iterations)):
        x = istft_vec(project(stft_vec(x, fft_size, hopsamp), mag), fft_size, hopsamp)
    return x


def griffin_lim_vec_f32(mag, fft_size, hopsamp, iterations, x0):
    """The same loop with every array and every FFT in float32 (scipy's pocketfft keeps single
    precision): what the reference's algorithm yields when numpy itself computes in the GPU's
    precision.  Used as a yardstick only: where Griffin-Lim is locally unstable (rounding errors
    grow about tenfold per 20 iterations at one spot) NO float32 implementation stays within
    1e-3 of the float64 reference, and the GPU result is then held to this one's error instead."""
    import scipy.fft as sf
    N, H = int(fft_size), int(hopsamp)
    mag = np.asarray(mag, dtype=np.float32)
    T = mag.shape[0]
    w = np.hanning(N).astype(np.float32)
    x = np.asarray(x0, dtype=np.float32)
    idx = (np.arange(T) * H)[:, None] + np.arange(N)[None, :]
    for _ in range(int(iterations)):
        S = sf.rfft(w[None, :] * x[idx], axis=1)
        a = np.abs(S)
        zero = a == 0
        a[zero] = 1.0
        P = (S / a) * mag
        P[zero] = mag[zero]                      # np.angle(0) == 0
        frames = sf.irfft(P, n=N, axis=1).astype(np.float32) * w[None, :]
        x = np.zeros(T * H + N, dtype=np.float32)
        for n in range(T):
            x[n * H:n * H + N] += frames[n]
    return x


# --------------------------------------------------------------------------
# Variants the reference does NOT have.  PARITY UNPINNED: there is no
# reference code or fixture for them; these are float64 statements of the textbook
# algorithms (Griffin & Lim 1984 least-squares ISTFT; Perraudin et al. 2013 fast
# Griffin-Lim written in the signal domain) used to check the HIP options.
# --------------------------------------------------------------------------
def window_sum(n_frames, fft_size, hopsamp):
    w2 = np.hanning(int(fft_size)) ** 2
    acc = np.zeros(int(n_frames) * int(hopsamp) + int(fft_size))
    for n in range(int(n_frames)):
        acc[n * hopsamp:n * hopsamp + fft_size] += w2
    return acc


def istft_vec_normalised(X, fft_size, hopsamp):
    y = istft_vec(X, fft_size, hopsamp)
    ws = window_sum(X.shape[0], fft_size, hopsamp)
    return np.where(ws > 1e-20, y / np.where(ws > 1e-20, ws, 1.0), y)


def griffin_lim_variant(mag, fft_size, hopsamp, iterations, x0, normalize=False, momentum=0.0):
    inv = istft_vec_normalised if normalize else istft_vec
    t = np.asarray(x0, dtype=np.float64)
    prev = None
    for it in range(int(iterations)):
        v = inv(project(stft_vec(t, fft_size, hopsamp), mag), fft_size, hopsamp)
        if momentum and prev is not None and it != iterations - 1:
            t = v + momentum * (v - prev)
        else:
            t = v
        prev = v
    return t


# --------------------------------------------------------------------------
# N3  forward analysis, data.py:100-116.  PARITY UNPINNED: librosa is absent (see the header);
# this is librosa.stft's documented algorithm (center=True, window='hann' periodic,
# hop_length = n_fft // 4 by default, pad_mode 'reflect' before 0.10 / 'constant' from 0.10).
# --------------------------------------------------------------------------
def stft_librosa_doc(y, n_fft, hop=None, pad_mode="reflect"):
    n_fft = int(n_fft)
    hop = n_fft // 4 if hop is None else int(hop)
    y = np.asarray(y, dtype=np.float64)
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)
    n_frames = 1 + len(y) // hop
    idx = (np.arange(n_frames) * hop)[:, None] + np.arange(n_fft)[None, :]
    return np.fft.rfft(w[None, :] * yp[idx], axis=1)          # (T, F), rows = frames


def audio_to_logmel(y, filterbank, n_fft, hop=None, pad_mode="reflect", compression_factor=1.0):
    """data.py:110-114: log(clip(filterbank @ |stft|, 0.01) * compression_factor) -> (n_mels, T)."""
    mag = np.abs(stft_librosa_doc(y, n_fft, hop, pad_mode))
    mel = np.dot(np.asarray(filterbank, dtype=np.float64), mag.T)
    return np.log(np.clip(mel, a_min=0.01, a_max=None) * compression_factor), mag


def spectral_convergence(x, mag, fft_size, hopsamp, fit_gain=True):
    """|| |STFT(x)|/c - mag ||_F / || mag ||_F  (SURVEY 8d parity gate).  The
    reference ISTFT applies no window-sum normalisation, so x carries a gain of
    about sum(w^2)/hop; c is the least-squares scalar that removes it."""
    S = np.abs(stft_vec(x, fft_size, hopsamp))
    mag = np.asarray(mag, dtype=np.float64)
    c = float(np.sum(S * mag) / max(np.sum(S * S), 1e-300)) if fit_gain else 1.0
    return float(np.linalg.norm(S * c - mag) / np.linalg.norm(mag))


# --------------------------------------------------------------------------
# a-1  librosa.filters.mel call site     griffin_lim.py:13-14   (UNPINNED)
# --------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_part = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_part, mels)


def _mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank_slaney(sr, n_fft, n_mels=80, fmin=0.0, fmax=None):
    """(n_mels, 1+n_fft//2) float32 triangular filterbank, Slaney mel scale,
    Slaney area normalisation -- librosa.filters.mel(htk=False, norm='slaney')
    as documented.  PARITY UNPINNED (see module docstring)."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, sr / 2.0, n_bins)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


# --------------------------------------------------------------------------
# a-2  save_mel_to_wav (everything but the file write)   griffin_lim.py:17-35
# --------------------------------------------------------------------------
def mel_to_linear(filterbank, mel):
    """griffin_lim.py:18-19: np.dot(filterbank.T, mel).T -> (T, F)."""
    return np.dot(filterbank.T, mel).T


def peak_normalise(x):
    """griffin_lim.py:27-29: divide by max|x| only when it exceeds 1."""
    peak = np.max(np.abs(x))
    if peak > 1.0:
        return x / peak
    return x


# --------------------------------------------------------------------------
# a-7  save_audio_to_file sample conversion   audio_utilities.py:202-210
# --------------------------------------------------------------------------
def pcm16_from_float(x):
    """assert max|x| <= 1; x /= max|x|; x *= 32767; int(round(.)) per sample.
    Python's round() is round-half-to-even == np.rint."""
    x = np.array(x, dtype=np.float64, copy=True)
    peak = np.max(np.abs(x))
    assert peak <= 1.0, 'Input audio value is out of range. Should be in the range [-1.0, 1.0].'
    x /= peak
    x *= 32767.0
    return np.rint(x).astype('<i2')


# ---- N3: the sample-rate conversion inside librosa.load (data.py:101) -----------------------------------
# TEST INFRASTRUCTURE (like the rest of this file).  Neither librosa nor resampy is in /root/reference or in this
# container: PARITY UNPINNED.  What is restated is the published algorithm of resampy 0.2.x, the resampler behind
# librosa.load's default res_type 'kaiser_best' in the librosa 0.6 / 0.7 of the reference's Python 3.7 environment
# (venv/pyvenv.cfg): resampy.filters.sinc_window for the filter, resampy.interpn.resample_f for the interpolation,
# librosa.core.resample for the output length (ceil) and the zero padding (util.fix_length).
RS_ZEROS, RS_PRECISION = 64, 9
RS_BETA, RS_ROLLOFF = 14.769656459379492, 0.9475937167399596


def kaiser_best_filter():
    """(interp_win, num_table): right wing of the windowed sinc, 2**precision points per zero crossing."""
    num_bits = 2 ** RS_PRECISION
    n = num_bits * RS_ZEROS
    sinc_win = RS_ROLLOFF * np.sinc(RS_ROLLOFF * np.linspace(0, RS_ZEROS, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, RS_BETA)[n:]
    return taper * sinc_win, num_bits


def resample_kaiser_best(x, sr_in, sr_out):
    """librosa.resample(x, sr_in, sr_out) for a 1-D float32 signal, resampy's loop order and float32 running sum."""
    x = np.asarray(x, dtype=np.float32)
    ratio = float(sr_out) / float(sr_in)
    n_out = int(np.ceil(x.shape[0] * ratio))
    n_calc = min(int(x.shape[0] * ratio), n_out)
    interp_win, num_table = kaiser_best_filter()
    if ratio < 1:
        interp_win = interp_win * ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * num_table)
    nwin = interp_win.shape[0]
    n_orig = x.shape[0]
    y = np.zeros(n_out, dtype=np.float32)
    for t in range(n_calc):
        time_register = t * time_increment
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        acc = np.float32(0.0)
        for i in range(i_max):
            weight = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step]
            acc = np.float32(np.float64(acc) + weight * np.float64(x[n - i]))
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        for k in range(k_max):
            weight = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step]
            acc = np.float32(np.float64(acc) + weight * np.float64(x[n + k + 1]))
        y[t] = acc
    return y
