#!/usr/bin/env python3
"""Lane-level numpy model of the fused wave-per-frame kernel's data movement
(gl_fft1024.h, gl_fast2048.hip), used to validate the index maps on the CPU before the HIP code
runs on a GPU -- TEST INFRASTRUCTURE ONLY (never imported by the product).

Model: one 64-lane wave per frame; M = N/2 complex points, R = M/64 per lane.
  N=2048: M=1024 = 16 (pass 1, over n2) x 8 (pass 2, over b) x 8 (pass 3, over a)
     n = a + 8 b + 64 n2          k = k2 + 16 d + 128 c
Each "lane" below is a row of a [64][R] complex array.
"""
import numpy as np


def W(n, k):
    return np.exp(-2j * np.pi * k / n)


def forward_2048(z):
    """z: [1024] complex (packed even/odd windowed frame) -> ZA[64][8], ZB[64][8]
    with ZA[l][c] = Z[l + 128 c], ZB[l][c] = Z[(128-l) + 128 c] (lane 0: Z[64 + 128 c])."""
    M = 1024
    # pass 1: lane n1, regs n2
    reg = np.zeros((64, 16), complex)
    for l in range(64):
        for j in range(16):
            reg[l, j] = z[l + 64 * j]
    y1 = np.zeros((64, 16), complex)
    for l in range(64):
        for k2 in range(16):
            y1[l, k2] = sum(reg[l, n2] * W(16, n2 * k2) for n2 in range(16)) * W(1024, l * k2)
    # exchange 1: lds1[k2*72 + n1]
    lds = np.zeros(16 * 72, complex)
    for l in range(64):
        for k2 in range(16):
            lds[k2 * 72 + l] = y1[l, k2]
    # pass 2: lane l2 = a + 8 g; butterflies k2 = g + 8 h; regs [h][b]
    y2 = np.zeros((64, 2, 8), complex)
    for l in range(64):
        a, g = l & 7, l >> 3
        for h in range(2):
            v = [lds[(g + 8 * h) * 72 + a + 8 * b] for b in range(8)]
            for d in range(8):
                y2[l, h, d] = sum(v[b] * W(8, b * d) for b in range(8)) * W(64, a * d)
    # exchange 2: lds2[a*130 + 16 d + k2]
    lds = np.zeros(8 * 130, complex)
    for l in range(64):
        a, g = l & 7, l >> 3
        for h in range(2):
            for d in range(8):
                lds[a * 130 + 16 * d + g + 8 * h] = y2[l, h, d]
    # pass 3: lane l3: butterfly A k_lo = l3, B k_lo = 128 - l3 (lane 0: 64)
    ZA = np.zeros((64, 8), complex)
    ZB = np.zeros((64, 8), complex)
    for l in range(64):
        for which, klo in ((0, l), (1, 128 - l if l else 64)):
            v = [lds[a * 130 + klo] for a in range(8)]
            out = [sum(v[a] * W(8, a * c) for a in range(8)) for c in range(8)]
            (ZA if which == 0 else ZB)[l] = out
    return ZA, ZB


def inverse_2048(ZA, ZB):
    """Exact reverse flow; returns z[1024] = IFFT (unnormalised: multiply by 1/1024 outside)."""
    lds = np.zeros(8 * 130, complex)
    for l in range(64):
        for which, klo in ((0, l), (1, 128 - l if l else 64)):
            src = ZA[l] if which == 0 else ZB[l]
            for a in range(8):
                lds[a * 130 + klo] = sum(src[c] * np.conj(W(8, a * c)) for c in range(8))
    y2 = np.zeros((64, 2, 8), complex)
    for l in range(64):
        a, g = l & 7, l >> 3
        for h in range(2):
            u = [lds[a * 130 + 16 * d + g + 8 * h] * np.conj(W(64, a * d)) for d in range(8)]
            for b in range(8):
                y2[l, h, b] = sum(u[d] * np.conj(W(8, b * d)) for d in range(8))
    lds = np.zeros(16 * 72, complex)
    for l in range(64):
        a, g = l & 7, l >> 3
        for h in range(2):
            for b in range(8):
                lds[(g + 8 * h) * 72 + a + 8 * b] = y2[l, h, b]
    z = np.zeros(1024, complex)
    for l in range(64):
        v = [lds[k2 * 72 + l] * np.conj(W(1024, l * k2)) for k2 in range(16)]
        for n2 in range(16):
            z[l + 64 * n2] = sum(v[k2] * np.conj(W(16, n2 * k2)) for k2 in range(16))
    return z


def pair_fwd(zk, zm, w):
    A = 0.5 * (zk + np.conj(zm))
    Zo = -0.5j * (zk - np.conj(zm))
    Bv = w * Zo
    return A + Bv, np.conj(A - Bv)


def pair_inv(Xk, Xmk, w):
    A = 0.5 * (Xk + np.conj(Xmk))
    Bv = 0.5 * (Xk - np.conj(Xmk))
    Zo = np.conj(w) * Bv
    return A + 1j * Zo, np.conj(A) + 1j * np.conj(Zo)


def project(X, m):
    r = abs(X)
    return X * (m / r) if r > 0 else complex(m, 0.0)


def middle_2048(ZA, ZB, mag):
    """rfft post-process + magnitude projection + irfft pre-process, all lane-local.
    mag: [1025].  Returns Z'A, Z'B and the rfft spectrum X[1025] (for checking)."""
    M = 1024
    X = np.zeros(M + 1, complex)
    ZA2 = np.zeros_like(ZA)
    ZB2 = np.zeros_like(ZB)
    for l in range(64):
        if l:
            P = [(0, c) for c in range(8)]               # (which, idx)
            Q = [(1, 7 - c) for c in range(8)]
            kP = [l + 128 * c for c in range(8)]
            extra = None
        else:
            P = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3)]
            Q = [(0, 0), (0, 7), (0, 6), (0, 5), (1, 7), (1, 6), (1, 5), (1, 4)]
            kP = [0, 128, 256, 384, 64, 192, 320, 448]
            extra = ((0, 4), (0, 4), 512)
        units = list(zip(P, Q, kP)) + ([extra] if extra else [])
        for (pw, pi), (qw, qi), k in units:
            src = (ZA, ZB)
            zk, zm = src[pw][l, pi], src[qw][l, qi]
            w = W(2048, k)
            Xk, Xmk = pair_fwd(zk, zm, w)
            X[k], X[M - k] = Xk, Xmk
            Xk, Xmk = project(Xk, mag[k]), project(Xmk, mag[M - k])
            z1, z2 = pair_inv(Xk, Xmk, w)
            dst = (ZA2, ZB2)
            dst[qw][l, qi] = z2
            dst[pw][l, pi] = z1       # P written last: for self pairs z1 == z2
    return ZA2, ZB2, X


def main():
    rng = np.random.default_rng(0)
    N, M = 2048, 1024
    win = np.hanning(N)
    x = rng.standard_normal(N)
    xw = win * x
    z = xw[0::2] + 1j * xw[1::2]
    ZA, ZB = forward_2048(z)
    Zref = np.fft.fft(z)
    for l in range(64):
        for c in range(8):
            assert abs(ZA[l, c] - Zref[l + 128 * c]) < 1e-9
            assert abs(ZB[l, c] - Zref[(128 - l if l else 64) + 128 * c]) < 1e-9
    print("forward map ok")
    zr = inverse_2048(ZA, ZB) / M
    assert np.max(np.abs(zr - z)) < 1e-12
    print("inverse map ok")
    mag = np.abs(rng.standard_normal(M + 1))
    ZA2, ZB2, X = middle_2048(ZA, ZB, mag)
    Xref = np.fft.rfft(xw)
    assert np.max(np.abs(X - Xref)) < 1e-9, np.max(np.abs(X - Xref))
    print("rfft post-process ok")
    z2 = inverse_2048(ZA2, ZB2) / M
    y = np.empty(N)
    y[0::2], y[1::2] = z2.real, z2.imag
    yref = np.fft.irfft(mag * np.exp(1j * np.angle(Xref)))
    assert np.max(np.abs(y - yref)) < 1e-9, np.max(np.abs(y - yref))
    assert np.max(np.abs(z2.imag - yref[1::2])) < 1e-9
    print("projection + irfft ok")


# ---------------------------------------------------------------------------------------
# n_fft = 1024: two real frames per 1024-point complex FFT (z = frame_a + i*frame_b).
# Same forward/inverse lane maps; the split needs no twiddle:
#   S = Z[k] + conj(Z[M-k]) = 2 Xa[k],  D = Z[k] - conj(Z[M-k]) = 2i Xb[k]
#   Z'[k] = S*sa + D*sb,  Z'[M-k] = conj(S*sa - D*sb),  sa = ma/|S|, sb = mb/|D|
# with the magnitude index min(k, M-k); zero S -> (ma, 0), zero D -> i*(mb, 0).
# ---------------------------------------------------------------------------------------
def middle_two_frames(ZA, ZB, mag_a, mag_b, a_zero=False, b_zero=False):
    M = 1024
    ZA2 = np.zeros_like(ZA)
    ZB2 = np.zeros_like(ZB)
    for l in range(64):
        if l:
            P = [(0, c) for c in range(8)]
            Q = [(1, 7 - c) for c in range(8)]
            kP = [l + 128 * c for c in range(8)]
            extra = None
        else:
            P = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3)]
            Q = [(0, 0), (0, 7), (0, 6), (0, 5), (1, 7), (1, 6), (1, 5), (1, 4)]
            kP = [0, 128, 256, 384, 64, 192, 320, 448]
            extra = ((0, 4), (0, 4), 512)
        units = list(zip(P, Q, kP)) + ([extra] if extra else [])
        for (pw, pi), (qw, qi), k in units:
            src = (ZA, ZB)
            zk, zm = src[pw][l, pi], src[qw][l, qi]
            kk = min(k, M - k)
            S = zk + np.conj(zm)
            D = zk - np.conj(zm)
            if a_zero:      # an all-zero input frame is forced to an exactly zero spectrum
                S = 0j      # (rounding noise of the partner frame must not set its phase)
            if b_zero:
                D = 0j
            Ya = S * (mag_a[kk] / abs(S)) if abs(S) > 0 else complex(mag_a[kk], 0)
            Yb = D * (mag_b[kk] / abs(D)) if abs(D) > 0 else complex(0, mag_b[kk])
            dst = (ZA2, ZB2)
            dst[qw][l, qi] = np.conj(Ya - Yb)
            dst[pw][l, pi] = Ya + Yb
    return ZA2, ZB2


def check_two_frames():
    rng = np.random.default_rng(3)
    N = 1024
    win = np.hanning(N)
    fa, fb = rng.standard_normal(N), rng.standard_normal(N)
    z = win * fa + 1j * (win * fb)
    ZA, ZB = forward_2048(z)
    mag_a, mag_b = np.abs(rng.standard_normal(513)), np.abs(rng.standard_normal(513))
    ZA2, ZB2 = middle_two_frames(ZA, ZB, mag_a, mag_b)
    z2 = inverse_2048(ZA2, ZB2) / 1024
    ya = np.fft.irfft(mag_a * np.exp(1j * np.angle(np.fft.rfft(win * fa))))
    yb = np.fft.irfft(mag_b * np.exp(1j * np.angle(np.fft.rfft(win * fb))))
    assert np.max(np.abs(z2.real - ya)) < 1e-9, np.max(np.abs(z2.real - ya))
    assert np.max(np.abs(z2.imag - yb)) < 1e-9, np.max(np.abs(z2.imag - yb))
    # zero input frame b -> phase 0 everywhere for b
    z = win * fa + 0j
    ZA, ZB = forward_2048(z)
    ZA2, ZB2 = middle_two_frames(ZA, ZB, mag_a, mag_b, b_zero=True)
    z2 = inverse_2048(ZA2, ZB2) / 1024
    assert np.max(np.abs(z2.imag - np.fft.irfft(mag_b.astype(complex)))) < 1e-9
    assert np.max(np.abs(z2.real - ya)) < 1e-9
    print("two-frames-per-FFT split (n_fft 1024) ok")


# --------------------------------------------------------------------------------------------
# LDS layouts of the four transposes (gl_fft1024.h: kE1P/kE1S, kE2Q/kE2S).  One side of every
# exchange moves runs of 8 contiguous complex values (16-byte instructions), the other side 8-byte
# accesses at permuted positions.  Checked here: (1) the permuted layout routes every value to the
# lane/register the plain transposes (n = a + 8b + 64 n2 maps above) send it to; (2) the worst bank
# conflict of every access pattern, per instruction kind and hardware lane group
# (MI355X_MICROARCH.md, LDS table), is at most 2-way.
# --------------------------------------------------------------------------------------------
E1P, E1S, E2Q, E2S = 10, 80, 10, 166


def _pi(l):
    return (l & 7) * E1P + (l >> 3)


def _rho(i):
    return (i & 15) * E2Q + (i >> 4)


def _klo(l):
    return 128 - l if l else 64


def check_exchange_routing():
    lanes = np.arange(64)
    a3, g3 = lanes & 7, lanes >> 3
    # exchange 1 forward: lane l writes z[k2] to (row k2, pi(l)); lane (a3,g3) reads rows g3+8h, a3*P+bb
    lds = np.full(8 * E2S, -1, dtype=np.int64)
    for k2 in range(16):
        lds[k2 * E1S + _pi(lanes)] = k2 * 64 + lanes            # tag = (k2, writer lane)
    for h in range(2):
        for bb in range(8):
            got = lds[(g3 + 8 * h) * E1S + a3 * E1P + bb]
            want = (g3 + 8 * h) * 64 + (a3 + 8 * bb)            # plain: ex[(g3+8h)*S + a3 + 8bb]
            assert np.array_equal(got, want)
    # exchange 2 forward: lane (a3,g3) writes z[8h+d] to (row a3, (g3+8h)*Q+d); lane l reads rho(l), rho(klo)
    lds[:] = -1
    for h in range(2):
        for d in range(8):
            lds[a3 * E2S + (g3 + 8 * h) * E2Q + d] = a3 * 128 + (16 * d + g3 + 8 * h)   # tag = (row, plain index)
    klo = np.array([_klo(l) for l in lanes])
    for a in range(8):
        assert np.array_equal(lds[a * E2S + _rho(lanes)], a * 128 + lanes)
        assert np.array_equal(lds[a * E2S + _rho(klo)], a * 128 + klo)
    assert 16 * E1S <= 8 * E2S and max(_rho(i) for i in range(128)) < E2S and max(_pi(l) for l in range(64)) < E1S
    return True


_R128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
         list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
_R128 = _R128 + [[32 + x for x in g] for g in _R128]
_R64 = [list(range(32)), list(range(32, 64))]
_W64 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
_W128 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def _worst_conflict(groups, addr_dw, ndw, nbanks):
    worst = 0
    for g in groups:
        per_bank = {}
        for l in g:
            for k in range(ndw):
                per_bank.setdefault((addr_dw(l) + k) % nbanks, set()).add(addr_dw(l) + k)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


def exchange_bank_conflicts():
    """-> {pattern: worst n-way conflict} (1 = conflict-free)."""
    a3 = lambda l: l & 7
    g3 = lambda l: l >> 3
    out = {}
    run1 = lambda h, c: (lambda l: 2 * ((g3(l) + 8 * h) * E1S + a3(l) * E1P + 2 * c))
    run2 = lambda h, c: (lambda l: 2 * (a3(l) * E2S + (g3(l) + 8 * h) * E2Q + 2 * c))
    hc = [(h, c) for h in range(2) for c in range(4)]
    out["ex1 fwd write b64"] = _worst_conflict(_W64, lambda l: 2 * _pi(l), 2, 32)
    out["ex1 fwd read b128"] = max(_worst_conflict(_R128, run1(h, c), 4, 64) for h, c in hc)
    out["ex1 inv write b128"] = max(_worst_conflict(_W128, run1(h, c), 4, 32) for h, c in hc)
    out["ex1 inv read b64"] = _worst_conflict(_R64, lambda l: 2 * _pi(l), 2, 64)
    out["ex2 fwd write b128"] = max(_worst_conflict(_W128, run2(h, c), 4, 32) for h, c in hc)
    out["ex2 fwd read b64"] = max(_worst_conflict(_R64, lambda l: 2 * _rho(l), 2, 64),
                                  _worst_conflict(_R64, lambda l: 2 * _rho(_klo(l)), 2, 64))
    out["ex2 inv write b64"] = max(_worst_conflict(_W64, lambda l: 2 * _rho(l), 2, 32),
                                   _worst_conflict(_W64, lambda l: 2 * _rho(_klo(l)), 2, 32))
    out["ex2 inv read b128"] = max(_worst_conflict(_R128, run2(h, c), 4, 64) for h, c in hc)
    return out


if __name__ == "__main__":
    check_exchange_routing()
    print(exchange_bank_conflicts())
    main()
    check_two_frames()
