#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference  --  TEST INFRASTRUCTURE.

Run in the build container only (needs /root/reference):

    python oracle/make_golden.py

It imports the reference's ``audio_utilities.py`` unmodified (it only needs
numpy/scipy), runs it on seeded inputs, checks the restatement in
``oracle/gl_oracle.py`` against it (<= 1e-12 relative) and stores inputs +
reference outputs as small .npz fixtures.  The reference itself never travels;
only these vectors do.  ``griffin_lim.py`` cannot be imported (librosa absent),
so save_mel_to_wav fixtures are built from its four numpy lines
(griffin_lim.py:18-29) wrapped around the reference
``reconstruct_signal_griffin_lim``, with the restated Slaney filterbank
(filterbank parity unpinned).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import audio_utilities as ref  # noqa: E402  (the reference, unmodified)
import gl_oracle as orc  # noqa: E402


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    den = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b))) / den


def check(name, a, b, tol=1e-12):
    e = relerr(a, b)
    status = "ok" if e <= tol else "MISMATCH"
    print(f"  restatement vs reference [{name}]: rel err {e:.3e} {status}")
    if e > tol:
        raise SystemExit(f"oracle restatement does not match the reference on {name}")


def smooth_mag(rng, T, F):
    """Speech-like smooth magnitude: exp of low-pass-filtered gaussian field."""
    u = rng.normal(-2.0, 1.5, size=(T + 8, F + 32))
    k_t = np.exp(-0.5 * (np.arange(-4, 5) / 1.5) ** 2)
    k_f = np.exp(-0.5 * (np.arange(-16, 17) / 6.0) ** 2)
    k_t /= k_t.sum()
    k_f /= k_f.sum()
    u = np.apply_along_axis(lambda r: np.convolve(r, k_f, mode="valid"), 1, u)
    u = np.apply_along_axis(lambda c: np.convolve(c, k_t, mode="valid"), 0, u)
    return np.ascontiguousarray(np.exp(u[:T, :F] * 2.0).astype(np.float32))


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    os.makedirs(GOLD, exist_ok=True)
    geoms = [(64, 16, 9), (256, 100, 7), (1024, 275, 12), (2048, 256, 12)]

    # ---- a-4 STFT ----------------------------------------------------------
    for (N, H, T) in geoms:
        rng = np.random.default_rng(1000 + N + H)
        x = rng.standard_normal(T * H + N)
        S = ref.stft_for_reconstruction(x, N, H)
        assert S.shape == (T, N // 2 + 1)
        check(f"stft {N}/{H}", orc.stft_loop(x, N, H), S)
        check(f"stft_vec {N}/{H}", orc.stft_vec(x, N, H), S)
        save(f"stft_{N}_{H}", x=x, fft_size=N, hop=H, S=S)

    # ---- a-5 ISTFT (non-zero imag at DC/Nyquist: irfft ignores them) --------
    for (N, H, T) in geoms:
        rng = np.random.default_rng(2000 + N + H)
        F = N // 2 + 1
        X = (rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F)))
        y = ref.istft_for_reconstruction(X, N, H)
        assert y.shape == (T * H + N,)
        assert np.all(y[-H:] == 0.0)
        check(f"istft {N}/{H}", orc.istft_loop(X, N, H), y)
        check(f"istft_vec {N}/{H}", orc.istft_vec(X, N, H), y)
        save(f"istft_{N}_{H}", X=X, fft_size=N, hop=H, y=y)

    # ---- a-3 Griffin-Lim, explicit x0 drawn the reference's way --------------
    gl_cases = [
        ("rand", 64, 16, 9), ("rand", 1024, 275, 12), ("rand", 2048, 256, 12),
        ("smooth", 1024, 275, 24), ("smooth", 2048, 256, 40),
    ]
    for kind, N, H, T in gl_cases:
        F = N // 2 + 1
        L = T * H + N
        rng = np.random.default_rng(3000 + N + H + T)
        if kind == "rand":
            mag = np.abs(rng.standard_normal((T, F))).astype(np.float32)
        else:
            mag = smooth_mag(rng, T, F)
        seed = 42
        np.random.seed(seed)
        x0 = np.random.randn(L)
        outs = {}
        for K in (1, 2, 5, 60):
            np.random.seed(seed)  # the reference draws x0 from the global legacy RNG
            y = ref.reconstruct_signal_griffin_lim(mag, N, H, K)
            outs[f"y{K}"] = y
        check(f"gl {kind} {N}/{H} K=5", orc.griffin_lim_loop(mag, N, H, 5, x0=x0), outs["y5"])
        check(f"gl_vec {kind} {N}/{H} K=60", orc.griffin_lim_vec(mag, N, H, 60, x0), outs["y60"], tol=1e-9)
        np.random.seed(seed)
        check(f"gl rng-path {kind} {N}/{H} K=2", orc.griffin_lim_loop(mag, N, H, 2), outs["y2"])
        save(f"gl_{kind}_{N}_{H}", mag=mag, x0=x0, seed=seed, fft_size=N, hop=H, **outs)

    # ---- F6: exact-zero bins / all-zero frame -------------------------------
    N, H, T = 256, 64, 10
    F = N // 2 + 1
    L = T * H + N
    rng = np.random.default_rng(77)
    mag = np.abs(rng.standard_normal((T, F))).astype(np.float32)
    mag[3, :] = 0.0
    mag[:, 5] = 0.0
    mag[:, F - 1] = 0.0
    x0 = rng.standard_normal(L)
    x0[:N + H] = 0.0  # frames 0 and 1 see all-zero input -> S == 0 -> angle 0 -> phase 1+0j
    outs = {}
    for K in (1, 3):
        outs[f"y{K}"] = _gl_with_x0(mag, N, H, K, x0)
    check("gl zero-bins K=3", orc.griffin_lim_loop(mag, N, H, 3, x0=x0), outs["y3"])
    save("gl_zero_bins", mag=mag, x0=x0, fft_size=N, hop=H, **outs)

    # ---- a-2: save_mel_to_wav body at full config-1 size ----------------------
    # griffin_lim.py:18-29 around the reference GL, params.py defaults
    # (n_fft 1024, hop 275, 80 mels, 22050 Hz, fmin 125, fmax 7600), 60 iterations,
    # T = 641 (~8 s), np.random.seed(42) like infer.py:34 -> data.py:22-26.
    N, H, T, K = 1024, 275, 641, 60
    rng = np.random.default_rng(1234)
    fb = orc.mel_filterbank_slaney(22050, N, 80, 125, 7600)
    u = rng.normal(-2.0, 1.5, size=(80 + 8, T + 4))
    kf = np.exp(-0.5 * (np.arange(-4, 5) / 1.6) ** 2); kf /= kf.sum()
    kt = np.exp(-0.5 * (np.arange(-2, 3) / 0.8) ** 2); kt /= kt.sum()
    u = np.apply_along_axis(lambda c: np.convolve(c, kf, mode="valid"), 0, u)
    u = np.apply_along_axis(lambda r: np.convolve(r, kt, mode="valid"), 1, u)
    mel = np.exp(u[:80, :T]).astype(np.float32)
    lin = np.dot(fb.T, mel)              # griffin_lim.py:18
    stft_modified = lin.T                # griffin_lim.py:19
    np.random.seed(42)
    y = ref.reconstruct_signal_griffin_lim(stft_modified, N, H, K)   # griffin_lim.py:23-24
    peak = np.max(abs(y))                # griffin_lim.py:27
    y_norm = y / peak if peak > 1.0 else y
    pcm = orc.pcm16_from_float(y_norm)
    save("save_mel_to_wav_cfg1", mel=mel, fft_size=N, hop=H, iters=K, seed=42,
         y=y.astype(np.float32), peak=peak, pcm=pcm)

    # ---- mel inversion only --------------------------------------------------
    rng = np.random.default_rng(5)
    basis = np.abs(rng.standard_normal((1025, 80))).astype(np.float32)
    mel = np.abs(rng.standard_normal((80, 37))).astype(np.float32)
    save("mel_inv_2048", basis=basis, mel=mel, mag=np.dot(basis, mel).T)

    # ---- a-7 known answers (hand-built; reference dies at tostring on py3.10) ---
    x = np.array([0.0, 0.5, -1.0, 1.0 / 65534.0, 3.0 / 65534.0, -1.0 / 65534.0, 0.25, 1.0])
    # peak = 1 -> x*32767: 0, 16383.5 -> 16384 (half to even), -32767, 0.5 -> 0,
    # 1.5 -> 2, -0.5 -> -0, 8191.75 -> 8192, 32767
    expect = np.array([0, 16384, -32767, 0, 2, 0, 8192, 32767], dtype="<i2")
    got = orc.pcm16_from_float(x)
    assert np.array_equal(got, expect), (got, expect)
    # cross-check against the reference's own per-sample loop (lines 202-210) run inline
    xr = x.copy(); xm = np.max(abs(xr)); xr /= xm; xr *= 32767.0
    loop = np.array([int(round(v)) for v in xr], dtype="<i2")
    assert np.array_equal(loop, expect)
    save("pcm16_known", x=x, pcm=expect)
    # a-7 on realistic data: fp32 samples (what the GPU holds) through the reference's own
    # arithmetic (lines 202-210 inline: float64 /=, *=, per-sample int(round()))
    rng = np.random.default_rng(77)
    xf = (rng.standard_normal(40000) * 0.2).astype(np.float32)
    xf[123] = 0.0
    xr = xf.astype(np.float64); xm = np.max(abs(xr)); xr /= xm; xr *= 32767.0
    loop = np.array([int(round(v)) for v in xr], dtype="<i2")
    assert np.array_equal(loop, orc.pcm16_from_float(xf.astype(np.float64)))
    save("pcm16_random", x=xf, pcm=loop)

    # ---- N1: infer.py:24-29 glue, batch of 3 (log-mel in -> samples and PCM out) -----
    # mels = dynamic_range_decompression(y_pred_post) = exp(.)/1 (data.py:84-85, fp32 like torch);
    # mel = mels[b].T (infer.py:28); then the body of save_mel_to_wav around the reference GL.
    N, H, T, K, Bn = 1024, 275, 48, 12, 3
    L = T * H + N
    rng = np.random.default_rng(99)
    fb = orc.mel_filterbank_slaney(22050, N, 80, 125, 7600)
    u = rng.normal(-2.5, 1.2, size=(Bn, T + 2, 80 + 2))
    u = (u[:, :-2] + u[:, 1:-1] + u[:, 2:]) / 3.0
    u = (u[:, :, :-2] + u[:, :, 1:-1] + u[:, :, 2:]) / 3.0
    logmel = u.astype(np.float32)                       # (B, T, 80) = y_pred_post layout
    x0 = rng.standard_normal((Bn, L))
    ys, peaks, pcms = [], [], []
    for b in range(Bn):
        mel = np.exp(logmel[b]).T                       # fp32, (80, T)
        stft_modified = np.dot(fb.T, mel).T
        y = _gl_with_x0(stft_modified, N, H, K, x0[b])
        peak = np.max(abs(y))
        y_norm = y / peak if peak > 1.0 else y
        ys.append(y); peaks.append(peak); pcms.append(orc.pcm16_from_float(y_norm))
    save("infer_glue", logmel=logmel, x0=x0.astype(np.float32), fft_size=N, hop=H, iters=K,
         y=np.array(ys, dtype=np.float32), peak=np.array(peaks), pcm=np.array(pcms))
    print("all golden vectors written")


def _gl_with_x0(mag, N, H, K, x0):
    """Run the reference GL with a chosen x0 by substituting the one RNG draw."""
    saved = np.random.randn
    try:
        np.random.randn = lambda n: x0.copy()
        return ref.reconstruct_signal_griffin_lim(mag, N, H, K)
    finally:
        np.random.randn = saved


if __name__ == "__main__":
    main()
