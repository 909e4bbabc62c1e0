"""Mel filterbank used by griffin_lim.filterbank.

The reference builds it with ``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)``
(griffin_lim.py:13-14).  librosa is a third-party dependency that is neither
vendored nor version-pinned by the reference and is not installed here, so this
module implements librosa's documented default algorithm (Slaney mel scale,
``htk=False``; Slaney area normalisation, ``norm='slaney'``; float32 output).
Parity with the reference's actual librosa build is therefore UNPINNED; any other
basis can be passed explicitly to ``save_mel_to_wav(..., basis=...)``.
"""
import numpy as np

_F_SP = 200.0 / 3.0          # Hz per mel below the knee
_KNEE_HZ = 1000.0
_KNEE_MEL = _KNEE_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(freq):
    freq = np.asarray(freq, dtype=np.float64)
    lin = freq / _F_SP
    log = _KNEE_MEL + np.log(np.maximum(freq, _KNEE_HZ) / _KNEE_HZ) / _LOGSTEP
    return np.where(freq >= _KNEE_HZ, log, lin)


def mel_to_hz(mel):
    mel = np.asarray(mel, dtype=np.float64)
    lin = _F_SP * mel
    log = _KNEE_HZ * np.exp(_LOGSTEP * (np.maximum(mel, _KNEE_MEL) - _KNEE_MEL))
    return np.where(mel >= _KNEE_MEL, log, lin)


def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """Triangular mel filterbank, shape (n_mels, 1 + n_fft//2), float32."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + int(n_fft) // 2
    bin_hz = np.linspace(0.0, float(sr) / 2, n_bins)
    edges_hz = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), int(n_mels) + 2))
    widths = np.diff(edges_hz)
    offset = edges_hz[:, None] - bin_hz[None, :]          # (n_mels+2, n_bins)
    rising = -offset[:-2] / widths[:-1, None]
    falling = offset[2:] / widths[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling))
    tri *= (2.0 / (edges_hz[2:] - edges_hz[:-2]))[:, None]  # Slaney: unit area per band
    return tri.astype(np.float32)


def pinv_basis(filterbank):
    """(F, n_mels) Moore-Penrose pseudo-inverse of a (n_mels, F) filterbank: the
    alternate inversion basis named by the north-star; NOT what the reference uses
    (it multiplies by filterbank.T, griffin_lim.py:18)."""
    return np.linalg.pinv(np.asarray(filterbank, dtype=np.float64)).astype(np.float32)
