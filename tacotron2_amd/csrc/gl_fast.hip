// Fused wave-per-frame Griffin-Lim kernels for gfx950 (MI355X), n_fft = 2048 and 1024.
//
// One 64-lane wavefront owns a run of G consecutive frames of one utterance and
// keeps the whole per-frame pipeline on chip:
//
//   x ring -> window -> 1024-point complex FFT in registers (16 points per
//   lane: radix-16, radix-8, radix-8 with two LDS transposes) -> real-FFT
//   post-process + magnitude projection + inverse pre-process, lane-local ->
//   inverse FFT (mirror image) -> window -> overlap-add ring ->
//   each output sample written to HBM exactly once.
//
// Kernels, in the order of the file:
//   k_fast2048<MODE>     n_fft 2048, any hop; rings in LDS; also the single-shot STFT / ISTFT
//   k_gl2048_regring<HS> n_fft 2048, hop 128*HS: rings in registers, two frames in flight per wave
//                        (the benchmark kernel)
//   k_gl1024_pair / k_gl1024_pair2   n_fft 1024, any hop: two real frames per complex FFT, runs of frames per wave
//   k_gl1024_frames<REACH> + k_ola_frames   n_fft 1024, small problems: one frame pair per wave, state kept as frames
//   k_gl2048_frames<HS>  n_fft 2048, hop 256*k, one or two utterances: one frame per wave, state kept as frames
//   k_analysis2048       n_fft 2048: |centred STFT| of the forward analysis path (wav -> log-mel), one wave per frame
// The FFT core (dft16/dft8, twiddle1/2, ex1_*/ex2_* transposes, middle_project) is shared.
//
// The index decomposition n = a + 8b + 64 n2, k = k2 + 16 d + 128 c and the lane
// assignment of the last pass (lane l owns the radix-8 butterflies k_lo = l and
// k_lo = 128 - l) put every mirror pair (k, 1024-k) of the real-FFT split in ONE
// lane, so the projection needs no cross-lane traffic.  The maps are modelled and
// checked on the CPU in oracle/sim_fast2048.py.
//
// A wave is its own workgroup: LDS operations of one wave execute in order, so the
// transposes need no barriers.  Runs of one utterance overlap by N-H samples; each
// run writes the partial sums of its trailing N-H samples to a side buffer ("tail")
// that the consumer (the next iteration's ring fill, or the final fold kernel) adds
// back -- plain stores, no atomics, fixed summation order (frame order, like the
// reference's `x[i:i+N] += ...` loop), so results are bit-reproducible and
// independent of batch composition.
//
// Replaces audio_utilities.py:91-92 (frame*window, rfft), :181-183 (projection),
// :112-113 (irfft, window, overlap-add) per iteration of :178-187.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "gl_internal.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kN = 2048;           // n_fft
constexpr int kM = 1024;           // complex points
constexpr int kRing = kN;          // ring length (floats), power of two
constexpr int kEx1Stride = 72;     // exchange-1 row stride (complex), 64 + 8 pad
constexpr int kEx2Stride = 130;    // exchange-2 row stride (complex), 128 + 2 pad
constexpr int kScratch = 16 * kEx1Stride;   // complex elements (>= 8*kEx2Stride)
static_assert(kScratch >= 8 * kEx2Stride, "scratch too small");

// per-lane constant table, entry-major: tab[e*64 + lane]
constexpr int kTabWin = 0;         // 32 floats: w[2(l+64j)], w[2(l+64j)+1]
constexpr int kTabTw1 = 32;        // 15 complex: W1024^(l*k2), k2 = 1..15
constexpr int kTabTw2 = 62;        // 7 complex:  W64^(a*d),   d = 1..7, a = l&7
constexpr int kTabTwp = 76;        // 9 complex:  W2048^(kP[u]), u = 0..8
constexpr int kTabEntries = 94;

enum { MODE_GL = 0, MODE_STFT = 1, MODE_ISTFT = 2 };

// ---- complex helpers on v2f --------------------------------------------------------------
#include "gl_pk.h"   // cmul, cmulc, cmul_add, ... and the one-instruction packed forms

// LDS read.  -DGL_LDS_VOLATILE keeps the optimiser from pairing reads into ds_read2_b64 /
// ds_read2st64_b64 (half the rate of two ds_read_b64 per MI355X_MICROARCH.md); measured 4 % SLOWER
// on config 2 (0.185 vs 0.178 ms) because volatile also pins the reads' order, so it is off.
#ifdef GL_LDS_VOLATILE
typedef const volatile __attribute__((address_space(3))) v2f* lds_cvp;
__device__ __forceinline__ v2f lds_rd(const v2f* q) { return *(lds_cvp)q; }
#else
__device__ __forceinline__ v2f lds_rd(const v2f* q) { return *q; }
#endif
typedef float v4f_ __attribute__((ext_vector_type(4)));
// two table entries of a pair (TE layout) with one 16-byte read
__device__ __forceinline__ void lds_rd_pair(const v2f* q, v2f& a, v2f& b) {
    const v4f_ t = *reinterpret_cast<const v4f_*>(q);
    a = t.xy;
    b = t.zw;
}

template <bool INV>
__device__ __forceinline__ v2f ctw(v2f a, v2f w) { return INV ? cmulc(a, w) : cmul(a, w); }
__device__ __forceinline__ v2f mul_mi(v2f a) { v2f r; r.x = a.y; r.y = -a.x; return r; }   // -i*a
__device__ __forceinline__ v2f mul_pi(v2f a) { v2f r; r.x = -a.y; r.y = a.x; return r; }   // +i*a
__device__ __forceinline__ v2f cconj(v2f a) { v2f r; r.x = a.x; r.y = -a.y; return r; }

// by-value select: `c ? z[i] : z[j]` on array lvalues would become a select of ADDRESSES
// (dynamic register indexing); passing by value keeps it a v_cndmask on data
__device__ __forceinline__ v2f sel(bool c, v2f a, v2f b) { v2f r; r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; return r; }

// The DFT kernels below are written in packed operations: with r = W4 = -i (forward) or +i
// (inverse), a + r*b and a - r*b are one v_pk_add each, a multiplication by W8 = (1 + r)/sqrt 2
// is one v_pk_add (a + r*a) whose 1/sqrt 2 rides the next butterfly as a v_pk_fma.
template <bool INV> __device__ __forceinline__ v2f add_r(v2f a, v2f b) { return INV ? pk_addi(a, b) : pk_subi(a, b); }
template <bool INV> __device__ __forceinline__ v2f sub_r(v2f a, v2f b) { return INV ? pk_subi(a, b) : pk_addi(a, b); }
__device__ __forceinline__ v2f fma_s(v2f a, float s, v2f c) { return __builtin_elementwise_fma(a, (v2f){s, s}, c); }  // c + s*a
// a * (wx + i*wy) for a compile-time twiddle (conjugated when INV)
template <bool INV>
__device__ __forceinline__ v2f cmul_k(v2f a, float wx, float wy) {
    const float y = INV ? -wy : wy;
    return __builtin_elementwise_fma(a.yy, (v2f){-y, wx}, a.xx * (v2f){wx, y});
}

constexpr float kH = 0.70710678118654752440f;   // 1/sqrt 2

template <bool INV>
__device__ __forceinline__ void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
    v2f s0 = a0 + a2, s1 = a0 - a2, s2 = a1 + a3, s3 = a1 - a3;
    a0 = s0 + s2;
    a2 = s0 - s2;
    a1 = add_r<INV>(s1, s3);
    a3 = sub_r<INV>(s1, s3);
}

// natural-order in-place DFT-8 on v[0..7]  (26 packed operations)
template <bool INV>
__device__ __forceinline__ void dft8(v2f* v) {
    dft4<INV>(v[0], v[2], v[4], v[6]);   // E0..E3 in v0,v2,v4,v6
    dft4<INV>(v[1], v[3], v[5], v[7]);   // O0..O3 in v1,v3,v5,v7
    const v2f e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o2 = v[5];
    const v2f t1 = add_r<INV>(v[3], v[3]);   // O1 * W8^1 =  h * t1
    const v2f t3 = sub_r<INV>(v[7], v[7]);   // O3 * W8^3 = -h * t3
    v[0] = e0 + o0;             v[4] = e0 - o0;
    v[1] = fma_s(t1, kH, e1);   v[5] = fma_s(t1, -kH, e1);
    v[2] = add_r<INV>(e2, o2);  v[6] = sub_r<INV>(e2, o2);
    v[3] = fma_s(t3, -kH, e3);  v[7] = fma_s(t3, kH, e3);
}

// natural-order in-place DFT-16 on v[0..15]  (78 packed operations)
template <bool INV>
__device__ __forceinline__ void dft16(v2f* v) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;  // cos, sin(pi/8)
    // step 1: for rho = 0..3, DFT4 over m of x[rho + 4m] -> E_rho[q] kept in v[rho + 4q]
#pragma unroll
    for (int r = 0; r < 4; ++r) dft4<INV>(v[r], v[r + 4], v[r + 8], v[r + 12]);
    // steps 2+3: for q = 0..3, DFT4 over rho of E_rho[q] * W16^(rho*q) -> X[q + 4p]
    v2f out[16];
    {   // q = 0
        v2f a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        dft4<INV>(a0, a1, a2, a3);
        out[0] = a0; out[4] = a1; out[8] = a2; out[12] = a3;
    }
    {   // q = 1: twiddles 1, W16^1, W8^1, W16^3
        const v2f a1 = cmul_k<INV>(v[5], c1, -s1), a3 = cmul_k<INV>(v[7], s1, -c1);
        const v2f t2 = add_r<INV>(v[6], v[6]);                     // v6 * W8^1 = h * t2
        const v2f s0 = fma_s(t2, kH, v[4]), s1_ = fma_s(t2, -kH, v[4]);
        const v2f s2 = a1 + a3, s3 = a1 - a3;
        out[1] = s0 + s2; out[9] = s0 - s2; out[5] = add_r<INV>(s1_, s3); out[13] = sub_r<INV>(s1_, s3);
    }
    {   // q = 2: twiddles 1, W8^1, r, W8^3
        const v2f s0 = add_r<INV>(v[8], v[10]), s1_ = sub_r<INV>(v[8], v[10]);
        const v2f t1 = add_r<INV>(v[9], v[9]), t1r = sub_r<INV>(v[9], v[9]);       // a1 = h*t1,  r*a1 = -h*t1r
        const v2f t3 = sub_r<INV>(v[11], v[11]), t3r = add_r<INV>(v[11], v[11]);   // a3 = -h*t3, r*a3 = -h*t3r
        const v2f u2 = t1 - t3;      // (a1 + a3) / h
        const v2f u3 = t3r - t1r;    // r * (a1 - a3) / h
        out[2] = fma_s(u2, kH, s0);  out[10] = fma_s(u2, -kH, s0);
        out[6] = fma_s(u3, kH, s1_); out[14] = fma_s(u3, -kH, s1_);
    }
    {   // q = 3: twiddles 1, W16^3, W8^3, W16^9 = -W16^1
        const v2f a1 = cmul_k<INV>(v[13], s1, -c1), a3 = cmul_k<INV>(v[15], -c1, s1);
        const v2f t2 = sub_r<INV>(v[14], v[14]);                   // v14 * W8^3 = -h * t2
        const v2f s0 = fma_s(t2, -kH, v[12]), s1_ = fma_s(t2, kH, v[12]);
        const v2f s2 = a1 + a3, s3 = a1 - a3;
        out[3] = s0 + s2; out[11] = s0 - s2; out[7] = add_r<INV>(s1_, s3); out[15] = sub_r<INV>(s1_, s3);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = out[i];
}

// rfft split for the mirror pair (k, M-k): Z[k], Z[M-k] -> X[k], X[M-k]
__device__ __forceinline__ void pair_fwd(v2f zk, v2f zm, v2f w, v2f& Xk, v2f& Xmk) {
    v2f A, Zo;
    A.x = 0.5f * (zk.x + zm.x);  A.y = 0.5f * (zk.y - zm.y);
    Zo.x = 0.5f * (zk.y + zm.y); Zo.y = -0.5f * (zk.x - zm.x);
    v2f Bv = cmul(w, Zo);
    Xk = A + Bv;
    Xmk = cconj(A - Bv);
}
// inverse: X[k], X[M-k] -> Z[k], Z[M-k] of the packed inverse transform
__device__ __forceinline__ void pair_inv(v2f Xk, v2f Xmk, v2f w, v2f& zk, v2f& zm) {
    v2f A, Bv;
    A.x = 0.5f * (Xk.x + Xmk.x);  A.y = 0.5f * (Xk.y - Xmk.y);
    Bv.x = 0.5f * (Xk.x - Xmk.x); Bv.y = 0.5f * (Xk.y + Xmk.y);
    v2f Zo = cmulc(Bv, w);
    zk.x = A.x - Zo.y;  zk.y = A.y + Zo.x;
    zm.x = A.x + Zo.y;  zm.y = Zo.x - A.y;
}
#ifndef GL_PROJ_EPS
#define GL_PROJ_EPS 1
#endif
__device__ __forceinline__ v2f project(v2f X, float m) {
#if GL_PROJ_EPS
    // select-free form: nudging the real part by 2^-60 turns an exactly zero bin into (2^-60, 0), whose
    // projection is (mag, 0) = mag * exp(1j * angle(0)) (audio_utilities.py:181-183) without a compare;
    // a bin that large enough to matter (|X| > 2^-36 ~ 1.5e-11) does not see the nudge in fp32
    const v2f Xe = X + (v2f){0x1p-60f, 0.0f};
    const float r2 = fmaxf(fmaf(Xe.x, Xe.x, Xe.y * Xe.y), 0x1p-124f);
    const float sc = m * __builtin_amdgcn_rsqf(r2);
    return Xe * (v2f){sc, sc};
#else
    float2 r = gl_project(make_float2(X.x, X.y), m);
    v2f o = {r.x, r.y};
    return o;
#endif
}

struct FastArgs {
    const float* x_in;       // [B][L]   (GL, STFT)
    long long x_stride;      // samples between utterances of x_in
    const float* tail_in;    // [B][nruns][N-H] or null
    int g_in;                // frames per run used by the producer of tail_in
    int nruns_in;
    const float* mag;        // [B][T][F]          (GL)
    const float2* S_in;      // [B][T][F] complex  (ISTFT)
    float2* S_out;           // [B][T][F] complex  (STFT)
    float* x_out;            // [B][L]             (GL, ISTFT)
    float* tail_out;         // [B][nruns][N-H]
    const float* tab;        // per-lane constants
    int H, T, G, nruns;
    long long L;
};

// x sample s of utterance row `xr`, with the producer's run tails folded in
__device__ __forceinline__ float fetch_x(const float* __restrict__ xr, const float* __restrict__ tr, long long s,
                                         int g_in_h, int nruns_in, int ovl) {
    float v = xr[s];
    if (tr != nullptr) {
        int k = (int)(s / g_in_h);
        if (k > nruns_in - 1) k = nruns_in - 1;
        int i = (int)(s - (long long)k * g_in_h);
        if (k >= 1 && i < ovl) v += tr[(long long)(k - 1) * ovl + i];
    }
    return v;
}

template <int MODE>
__global__ void __launch_bounds__(64) k_fast2048(FastArgs p) {
    __shared__ __attribute__((aligned(16))) float xring[kRing];
    __shared__ __attribute__((aligned(16))) float acc[kRing];
    __shared__ __attribute__((aligned(16))) v2f ex[kScratch];

    const int l = threadIdx.x;
    const int b = blockIdx.y;
    const int run = blockIdx.x;
    const int H = p.H, T = p.T;
    const int f0 = run * p.G;
    const int f1 = min(f0 + p.G, T);
    const int ovl = kN - H;                       // samples shared by neighbouring runs
    const int a3 = l & 7, g3 = l >> 3;            // pass-2 lane coordinates
    const int kloB = l ? 128 - l : 64;            // pass-3 second butterfly

    // ---- per-lane constants ----------------------------------------------------------
    const float* tab = p.tab + l;
    float win[32];
    v2f tw1[16], tw2[8], twp[9];
#pragma unroll
    for (int i = 0; i < 32; ++i) win[i] = tab[(kTabWin + i) * 64];
#pragma unroll
    for (int i = 1; i < 16; ++i) { tw1[i].x = tab[(kTabTw1 + 2 * (i - 1)) * 64]; tw1[i].y = tab[(kTabTw1 + 2 * (i - 1) + 1) * 64]; }
#pragma unroll
    for (int i = 1; i < 8; ++i) { tw2[i].x = tab[(kTabTw2 + 2 * (i - 1)) * 64]; tw2[i].y = tab[(kTabTw2 + 2 * (i - 1) + 1) * 64]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) { twp[i].x = tab[(kTabTwp + 2 * i) * 64]; twp[i].y = tab[(kTabTwp + 2 * i + 1) * 64]; }
    int kP[9];
#pragma unroll
    for (int c = 0; c < 8; ++c) kP[c] = l ? (l + 128 * c) : (c < 4 ? 128 * c : 64 + 128 * (c - 4));
    kP[8] = 512;

    const float* xr = nullptr;
    const float* tr = nullptr;
    const int g_in_h = p.g_in * H;
    if (MODE != MODE_ISTFT) {
        xr = p.x_in + (long long)b * p.x_stride;
        if (p.tail_in) tr = p.tail_in + (long long)b * p.nruns_in * ovl;
    }

    // ---- ring initialisation ----------------------------------------------------------------
    if (MODE != MODE_STFT) {
#pragma unroll
        for (int i = 0; i < kRing / 64; ++i) acc[l + 64 * i] = 0.0f;
    }
    if (MODE != MODE_ISTFT) {
        const long long s0 = (long long)f0 * H;
        for (int i = l; i < ovl; i += 64) xring[(int)((s0 + i) & (kRing - 1))] = fetch_x(xr, tr, s0 + i, g_in_h, p.nruns_in, ovl);
    }

    for (int n = f0; n < f1; ++n) {
        const long long sn = (long long)n * H;      // first sample of frame n
        v2f z[16];
        float m_lo[9], m_hi[9];

        if (MODE == MODE_GL) {
            const float* mrow = p.mag + ((long long)b * T + n) * (kM + 1);
#pragma unroll
            for (int u = 0; u < 9; ++u) { m_lo[u] = mrow[kP[u]]; m_hi[u] = mrow[kM - kP[u]]; }
        }

        if (MODE != MODE_ISTFT) {
            // append the H new samples [sn + N - H, sn + N) and read the frame
            for (int i = l; i < H; i += 64) {
                long long s = sn + ovl + i;
                xring[(int)(s & (kRing - 1))] = fetch_x(xr, tr, s, g_in_h, p.nruns_in, ovl);
            }
            __builtin_amdgcn_wave_barrier();
            const int base = (int)(sn & (kRing - 1));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int q = base + 2 * (l + 64 * j);
                z[j].x = xring[q & (kRing - 1)] * win[2 * j];
                z[j].y = xring[(q + 1) & (kRing - 1)] * win[2 * j + 1];
            }
            // ---- forward pass 1: DFT-16 over n2, twiddle W1024^(n1*k2) -------------------------
            dft16<false>(z);
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) z[k2] = cmul(z[k2], tw1[k2]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) ex[k2 * kEx1Stride + l] = z[k2];
            __builtin_amdgcn_wave_barrier();
            // ---- pass 2: lane (a, g): butterflies k2 = g + 8h over b, twiddle W64^(a*d) -----------
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) z[8 * h + bb] = ex[(g3 + 8 * h) * kEx1Stride + a3 + 8 * bb];
            dft8<false>(z);
            dft8<false>(z + 8);
#pragma unroll
            for (int d = 1; d < 8; ++d) { z[d] = cmul(z[d], tw2[d]); z[8 + d] = cmul(z[8 + d], tw2[d]); }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int d = 0; d < 8; ++d) ex[a3 * kEx2Stride + 16 * d + g3 + 8 * h] = z[8 * h + d];
            __builtin_amdgcn_wave_barrier();
            // ---- pass 3: butterflies k_lo = l (A) and 128-l (B) over a ------------------------------
#pragma unroll
            for (int a = 0; a < 8; ++a) { z[a] = ex[a * kEx2Stride + l]; z[8 + a] = ex[a * kEx2Stride + kloB]; }
            dft8<false>(z);
            dft8<false>(z + 8);
        }

        // ---- middle: 9 mirror-pair units, lane-local ---------------------------------------------------
        // lanes >= 1: P[c] = A[c], Q[c] = B[7-c]; lane 0: see oracle/sim_fast2048.py
        {
            const bool l0 = (l == 0);
            v2f P[9], Q[9];
            if (MODE != MODE_ISTFT) {
                P[0] = z[0];
                Q[0] = sel(l0, z[0], z[15]);
#pragma unroll
                for (int c = 1; c < 4; ++c) { P[c] = z[c]; Q[c] = sel(l0, z[8 - c], z[15 - c]); }
#pragma unroll
                for (int c = 4; c < 8; ++c) { P[c] = sel(l0, z[8 + c - 4], z[c]); Q[c] = sel(l0, z[8 + 11 - c], z[15 - c]); }
                P[8] = z[4];
                Q[8] = z[4];
            }
            v2f Pn[9], Qn[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                v2f Xk, Xmk;
                if (MODE != MODE_ISTFT) {
                    pair_fwd(P[u], Q[u], twp[u], Xk, Xmk);
                } else {
                    const float2* srow = p.S_in + ((long long)b * T + n) * (kM + 1);
                    float2 s1 = srow[kP[u]], s2 = srow[kM - kP[u]];
                    Xk.x = s1.x; Xk.y = s1.y; Xmk.x = s2.x; Xmk.y = s2.y;
                    if (u == 0 && l0) { Xk.y = 0.0f; Xmk.y = 0.0f; }   // irfft ignores imag(DC), imag(Nyquist)
                }
                if (MODE == MODE_STFT) {
                    float2* orow = p.S_out + ((long long)b * T + n) * (kM + 1);
                    if (u < 8 || l0) {
                        orow[kP[u]] = make_float2(Xk.x, Xk.y);
                        orow[kM - kP[u]] = make_float2(Xmk.x, Xmk.y);
                    }
                } else {
                    if (MODE == MODE_GL) { Xk = project(Xk, m_lo[u]); Xmk = project(Xmk, m_hi[u]); }
                    pair_inv(Xk, Xmk, twp[u], Pn[u], Qn[u]);
                }
            }
            if (MODE == MODE_STFT) continue;
            // scatter back to the A/B butterfly registers
            z[0] = Pn[0];
#pragma unroll
            for (int c = 1; c < 4; ++c) z[c] = Pn[c];
            z[4] = sel(l0, Pn[8], Pn[4]);
#pragma unroll
            for (int c = 5; c < 8; ++c) z[c] = sel(l0, Qn[8 - c], Pn[c]);
            // B[i], i = 0..7
#pragma unroll
            for (int i = 0; i < 4; ++i) z[8 + i] = sel(l0, Pn[4 + i], Qn[7 - i]);
#pragma unroll
            for (int i = 4; i < 8; ++i) z[8 + i] = sel(l0, Qn[11 - i], Qn[7 - i]);
        }

        // ---- inverse pass 3': inverse DFT-8 over c -> a, back through exchange 2 -------------------------
        dft8<true>(z);
        dft8<true>(z + 8);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int a = 0; a < 8; ++a) { ex[a * kEx2Stride + l] = z[a]; ex[a * kEx2Stride + kloB] = z[8 + a]; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int d = 0; d < 8; ++d) z[8 * h + d] = ex[a3 * kEx2Stride + 16 * d + g3 + 8 * h];
#pragma unroll
        for (int d = 1; d < 8; ++d) { z[d] = cmulc(z[d], tw2[d]); z[8 + d] = cmulc(z[8 + d], tw2[d]); }
        dft8<true>(z);
        dft8<true>(z + 8);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) ex[(g3 + 8 * h) * kEx1Stride + a3 + 8 * bb] = z[8 * h + bb];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) z[k2] = ex[k2 * kEx1Stride + l];
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2) z[k2] = cmulc(z[k2], tw1[k2]);
        dft16<true>(z);

        // ---- window, 1/M, overlap-add into the acc ring ----------------------------------------------------
        {
            const float sc = 1.0f / (float)kM;
            const int base = (int)(sn & (kRing - 1));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int q = base + 2 * (l + 64 * j);
                int q0 = q & (kRing - 1), q1 = (q + 1) & (kRing - 1);
                acc[q0] += (z[j].x * sc) * win[2 * j];
                acc[q1] += (z[j].y * sc) * win[2 * j + 1];
            }
            __builtin_amdgcn_wave_barrier();
            // samples [sn, sn+H) are complete: emit and clear
            float* orow = p.x_out + (long long)b * p.L;
            for (int i = l; i < H; i += 64) {
                int q = (int)((sn + i) & (kRing - 1));
                orow[sn + i] = acc[q];
                acc[q] = 0.0f;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    if (MODE != MODE_STFT) {
        // partial sums of the N-H samples after the run's last frame
        const long long se = (long long)f1 * H;
        if (f1 == T) {
            float* orow = p.x_out + (long long)b * p.L;
            for (int i = l; i < ovl; i += 64) orow[se + i] = acc[(int)((se + i) & (kRing - 1))];
            for (int i = l; i < H; i += 64) orow[se + ovl + i] = 0.0f;     // no frame covers the last H samples
        } else {
            float* trow = p.tail_out + ((long long)b * p.nruns + run) * ovl;
            for (int i = l; i < ovl; i += 64) trow[i] = acc[(int)((se + i) & (kRing - 1))];
        }
    }
}


// =====================================================================================
// Register-ring variant for hops that are a multiple of 128 samples (H = 128*HS).
//
// Lane l holds frame samples 2(l+64j), 2(l+64j)+1 in register j, so a hop of 128*HS
// samples is a shift by HS registers: the un-windowed input frame (xraw) and the
// overlap-add accumulator (accr) both live in registers and slide by HS rows per
// frame.  Per frame the wave loads HS new rows (prefetched one frame ahead), stores
// HS finished rows, and uses LDS for the four FFT transposes (10.4 KB per wave) and the
// window / split-twiddle constants (33 KB table shared by the workgroup); the FFT twiddles
// live in registers.  8 waves per CU (2 per SIMD, 242 VGPRs); the frame loop keeps two
// frames in flight per wave (GL_PIPE, see the loop).
// =====================================================================================
#ifndef GL_ABL
#define GL_ABL 0   // timing experiments only: 1 skip LDS transposes, 2 skip projection step, 4 constant twiddles, 16 skip the overlap-add of k_gl1024_pair2
#endif
#ifndef GL_SETPRIO
#define GL_SETPRIO 0   // experiment: raise the wave's issue priority while it feeds the LDS queue
#endif
__device__ __forceinline__ void prio_hi() { if (GL_SETPRIO) __builtin_amdgcn_s_setprio(GL_SETPRIO); }
__device__ __forceinline__ void prio_lo() { if (GL_SETPRIO) __builtin_amdgcn_s_setprio(0); }
// Transpose layouts of fft_forward_1024 / fft_inverse_1024.  Each exchange is written (or read) by a
// lane as runs of 8 contiguous complex values, so that side moves 16 bytes per instruction
// (4 x b128 instead of 8 x b64); the other side is 8-byte accesses at the permuted positions
// pi(l) = (l&7)*10 + (l>>3) (exchange 1, rows of 80) and rho(i) = (i&15)*10 + (i>>4) (exchange 2, rows
// of 166).  The pads (10 instead of 8, 166 instead of 160) make every access pattern at most 2-way
// bank-conflicting (checked per instruction kind and lane group in oracle/sim_fast2048.py).
constexpr int kE1P = 10, kE1S = 80;
constexpr int kE2Q = 10, kE2S = 166;
constexpr int kRScratch = 8 * kE2S;   // complex elements per wave (>= 16 * kE1S)
static_assert(kRScratch >= 16 * kE1S, "scratch too small");
__device__ __forceinline__ void lds_wr_pair(v2f* q, v2f a, v2f b) {
    *reinterpret_cast<v4f_*>(q) = (v4f_){a.x, a.y, b.x, b.y};
}

constexpr int kT2Win = 0;       // 16: (w[2(l+64j)], w[2(l+64j)+1])
constexpr int kT2Tw1 = 16;      // 15: W1024^(l*k2), k2 = 1..15     (every group starts on an even entry: pair reads)
constexpr int kT2Tw2 = 32;      // 7:  W64^(a*d), d = 1..7
constexpr int kT2Twp = 40;      // 9:  -i * W2048^(kP[u])
constexpr int kT2A16F = 50;     // 8:  the lane's 16 coefficients of the forward DFT-16 matrix (dft16_mfma)
constexpr int kT2A16I = 58;     // 8:  ... of the inverse
constexpr int kT2Entries = 66;   // entries the register-ring kernel copies to LDS (33 KB)
constexpr int kT2WinSc = 66;    // 16: the window / 1024 (n_fft-1024 kernel only; the 2048 kernel folds its scale into the magnitudes)
constexpr int kT2EntriesPair = 82;   // entries the n_fft-1024 kernel copies
constexpr int kT2Zero = 82;      // global-only: one row of zeros after the table (the "no tail" operand of load_row)
// Entries are stored in pairs, [e/2][lane][e&1]: a lane's entries 2p and 2p+1 are 16 contiguous bytes, so
// constants that are used together (two window values, two twiddles) arrive with one ds_read_b128
// (full LDS rate) instead of two reads.  tabl = smem + 2*lane; entry e of the lane is tabl[TE(e)].
#define TE(e) ((((e) >> 1) * 128) + ((e) & 1))

// Phase timing for kernel studies (-DGL_STAMPS): every phase boundary drains the LDS queue and
// reads the shader clock; lane 0 of every wave adds its per-phase totals to g_stamp at exit.
// The drain serialises what the normal build overlaps, so the sum exceeds the normal run time.
#ifdef GL_STAMPS
__device__ unsigned long long g_stamp[24];
struct Stamps {
    unsigned long long prev;
    unsigned acc[20];
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int i = 0; i < 20; ++i) acc[i] = 0;
        prev = now();
    }
    __device__ __forceinline__ static unsigned long long now() {
        unsigned long long t;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
        return t;
    }
    __device__ __forceinline__ void mark(int i) {
        unsigned long long t = now();
        acc[i] += (unsigned)(t - prev);
        prev = t;
    }
    __device__ __forceinline__ void flush(int lane) {
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 20; ++i) atomicAdd(&g_stamp[i], (unsigned long long)acc[i]);
            atomicAdd(&g_stamp[20], 1ull);
        }
    }
};
#else
struct Stamps {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int) {}
};
#endif

// Where the per-lane constants come from: the LDS copy of the table (entry e of the lane at tabl[TE(e)]) or,
// for a kernel that runs 2 waves per SIMD and has the registers, an array filled once per wave.
struct TabLds {
    const v2f* tabl;
    __device__ __forceinline__ v2f get(int e) const { return lds_rd(&tabl[TE(e)]); }
    __device__ __forceinline__ void pair(int e, v2f& a, v2f& b) const { lds_rd_pair(&tabl[TE(e)], a, b); }
};
struct TabReg {
    const v2f* r;
    __device__ __forceinline__ v2f get(int e) const { return r[e]; }
    __device__ __forceinline__ void pair(int e, v2f& a, v2f& b) const { a = r[e]; b = r[e + 1]; }
};

// the two FFT twiddle groups (entries kT2Tw1 .. kT2Twp-1) from registers, windows and split twiddles from LDS
struct TabMixed {
    const v2f* tabl;
    const v2f* r;        // r[e - kT2Tw1]
    __device__ __forceinline__ bool in_reg(int e) const { return e >= kT2Tw1 && e < kT2Twp; }
    __device__ __forceinline__ v2f get(int e) const { return in_reg(e) ? r[e - kT2Tw1] : lds_rd(&tabl[TE(e)]); }
    __device__ __forceinline__ void pair(int e, v2f& a, v2f& b) const {
        if (in_reg(e)) { a = r[e - kT2Tw1]; b = r[e + 1 - kT2Tw1]; }
        else lds_rd_pair(&tabl[TE(e)], a, b);
    }
};

// DFT-16 on the matrix cores.  The butterflies above cost 78 packed VALU operations per call and the fused
// kernels are bound by VALU issue while the MFMA pipe idles, so the radix-16 pass is also available as a dense
// real product C[32 x 64] = A[32 x 32] * B[32 x 64] on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains of 16
// terms): B row 2s / 2s+1 = re / im of input s (a lane's column is its own 16 inputs), A = the DFT matrix
// with output row i = component i&1 of output k2 = i>>1 (per-lane coefficients from the table).  One
// v_permlane32_swap per input turns (re, im) of the two lane halves into the B operands of the two 32-column
// blocks; one per accumulator register afterwards gives every lane all 32 rows of its own column, already as
// (re, im) register pairs.  32 MFMAs + 32 swaps + 4 table reads replace the 78 packed operations.
// MEASURED AND REJECTED (kept for the record, -DGL_MFMA16=1, parity-green): 0.226 ms per iteration against
// 0.155 ms with the butterflies on config 2 -- 64 dependent fp32 MFMAs per frame (2 chains of 16 per call,
// 64 cycles each) cost a wave more than the 156 packed operations they replace.
#ifndef GL_MFMA16
#define GL_MFMA16 0
#endif
typedef float f32x16_ __attribute__((ext_vector_type(16)));
template <bool INV, class TAB>
__device__ __forceinline__ void dft16_mfma(v2f* z, const TAB& tb) {
    f32x16_ c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c0[r] = 0.0f; c1[r] = 0.0f; }
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
        v2f a01, a23;
        tb.pair((INV ? kT2A16I : kT2A16F) + 2 * sp, a01, a23);
        const float av[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * sp + q;
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(z[s].x), __float_as_uint(z[s].y), false, false);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], __uint_as_float(sw[0]), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], __uint_as_float(sw[1]), c1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(c0[r]), __float_as_uint(c1[r]), false, false);
        c0[r] = __uint_as_float(sw[0]);
        c1[r] = __uint_as_float(sw[1]);
    }
    // accumulator register r of a lane now holds row (r&3) + 8(r>>2) (c0) / that + 4 (c1) of the lane's column
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k2 = (q & 1) + 4 * (q >> 1);
        z[k2] = (v2f){c0[2 * q], c0[2 * q + 1]};
        z[k2 + 2] = (v2f){c1[2 * q], c1[2 * q + 1]};
    }
}
template <bool INV, class TAB>
__device__ __forceinline__ void dft16_any(v2f* z, const TAB& tb) {
    if (GL_MFMA16) dft16_mfma<INV>(z, tb);
    else dft16<INV>(z);
}

#ifndef GL_TW2_REG
#define GL_TW2_REG 0   // 1: the seven W64 twiddles live in registers (callers pass them) instead of LDS reads
#endif
// ---- the 1024-point transform in phases -----------------------------------------------------------------
// (the composite fft_forward_1024 / fft_inverse_1024 below run them back to back; the software-pipelined
// register-ring loop interleaves the phases of two frames so that one frame's arithmetic covers the other's
// LDS round trips)
struct LanePos {
    int a3, g3, piL, rhoL, rhoK;
    __device__ __forceinline__ LanePos(int l, int kloB)
        : a3(l & 7), g3(l >> 3), piL((l & 7) * kE1P + (l >> 3)), rhoL((l & 15) * kE2Q + (l >> 4)),
          rhoK((kloB & 15) * kE2Q + (kloB >> 4)) {}
};
template <bool INV, class TAB>
__device__ __forceinline__ void twiddle1(v2f* z, const TAB& tb) {        // z[k2] *= W1024^(+-l*k2)
#pragma unroll
    for (int pp = 0; pp < 7; ++pp) {
        v2f wa, wb;
        tb.pair(kT2Tw1 + 2 * pp, wa, wb);
        z[2 * pp + 1] = ctw<INV>(z[2 * pp + 1], wa);
        z[2 * pp + 2] = ctw<INV>(z[2 * pp + 2], wb);
    }
    z[15] = ctw<INV>(z[15], tb.get(kT2Tw1 + 14));
}
template <bool INV, class TAB>
__device__ __forceinline__ void twiddle2(v2f* z, const TAB& tb, const v2f* w2r) {   // z[d], z[8+d] *= W64^(+-a*d)
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {       // d = 2pp+1, 2pp+2
        v2f wa, wb;
        if (w2r) { wa = w2r[2 * pp]; wb = w2r[pp < 3 ? 2 * pp + 1 : 2 * pp]; }
        else if (pp < 3) tb.pair(kT2Tw2 + 2 * pp, wa, wb);
        else wa = wb = tb.get(kT2Tw2 + 6);
        z[2 * pp + 1] = ctw<INV>(z[2 * pp + 1], wa);
        z[8 + 2 * pp + 1] = ctw<INV>(z[8 + 2 * pp + 1], wa);
        if (pp < 3) {
            z[2 * pp + 2] = ctw<INV>(z[2 * pp + 2], wb);
            z[8 + 2 * pp + 2] = ctw<INV>(z[8 + 2 * pp + 2], wb);
        }
    }
}
// exchange 1: register k2 <-> lane bits g3.  "cols" side: 16 x b64 at pi(l); "rows" side: 2 x 4 x b128
// GL_PIN_READS (experiment, off): bit mask of transposes (1 ex1 rows, 2 ex1 cols, 4 ex2 cols, 8 ex2 rows) whose
// reads get a scheduling fence behind them, so that all 16 are issued right after the writes instead of being
// sunk to their first use (the compiler sinks them to save registers).  Config 2: 1 and 2 -1 % (noise), 4 +1.7 %,
// 8 +0.7 %, all four +4 % (246 VGPRs, no spills): the fence costs other reorderings more than the earlier issue gains.
#ifndef GL_PIN_READS
#define GL_PIN_READS 0
#endif
#define GL_PIN(bit) do { if (GL_PIN_READS & (bit)) __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ void ex1_put_cols(const v2f* z, v2f* ex, const LanePos& lp) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) ex[k2 * kE1S + lp.piL] = z[k2];
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void ex1_get_rows(v2f* z, const v2f* ex, const LanePos& lp) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int bb = 0; bb < 8; bb += 2)
            lds_rd_pair(&ex[(lp.g3 + 8 * h) * kE1S + lp.a3 * kE1P + bb], z[8 * h + bb], z[8 * h + bb + 1]);
    GL_PIN(1);
}
__device__ __forceinline__ void ex1_put_rows(const v2f* z, v2f* ex, const LanePos& lp) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int bb = 0; bb < 8; bb += 2)
            lds_wr_pair(&ex[(lp.g3 + 8 * h) * kE1S + lp.a3 * kE1P + bb], z[8 * h + bb], z[8 * h + bb + 1]);
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void ex1_get_cols(v2f* z, const v2f* ex, const LanePos& lp) {
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) z[k2] = lds_rd(&ex[k2 * kE1S + lp.piL]);
    GL_PIN(2);
}
// exchange 2: register d <-> lane bits a3.  "rows" side: 2 x 4 x b128; "cols" side: 16 x b64 at rho(l), rho(128-l)
__device__ __forceinline__ void ex2_put_rows(const v2f* z, v2f* ex, const LanePos& lp) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int d = 0; d < 8; d += 2)
            lds_wr_pair(&ex[lp.a3 * kE2S + (lp.g3 + 8 * h) * kE2Q + d], z[8 * h + d], z[8 * h + d + 1]);
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void ex2_get_cols(v2f* z, const v2f* ex, const LanePos& lp) {
#pragma unroll
    for (int a = 0; a < 8; ++a) { z[a] = lds_rd(&ex[a * kE2S + lp.rhoL]); z[8 + a] = lds_rd(&ex[a * kE2S + lp.rhoK]); }
    GL_PIN(4);
}
__device__ __forceinline__ void ex2_put_cols(const v2f* z, v2f* ex, const LanePos& lp) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int a = 0; a < 8; ++a) { ex[a * kE2S + lp.rhoL] = z[a]; ex[a * kE2S + lp.rhoK] = z[8 + a]; }
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void ex2_get_rows(v2f* z, const v2f* ex, const LanePos& lp) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int d = 0; d < 8; d += 2)
            lds_rd_pair(&ex[lp.a3 * kE2S + (lp.g3 + 8 * h) * kE2Q + d], z[8 * h + d], z[8 * h + d + 1]);
    GL_PIN(8);
}

template <class TAB>
__device__ __forceinline__ void fft_forward_1024(v2f* z, v2f* ex, int l, int a3, int g3, int kloB,
                                                 const TAB& tb, Stamps& st, const v2f* w2r = nullptr) {
    const LanePos lp(l, kloB);
    dft16_any<false>(z, tb);
    twiddle1<false>(z, tb);
    st.mark(1);
    if (!(GL_ABL & 1)) {
        prio_hi();
        ex1_put_cols(z, ex, lp);
        ex1_get_rows(z, ex, lp);
        prio_lo();
    }
    st.mark(2);
    dft8<false>(z);
    dft8<false>(z + 8);
    twiddle2<false>(z, tb, w2r);
    st.mark(3);
    if (!(GL_ABL & 1)) {
        prio_hi();
        ex2_put_rows(z, ex, lp);
        ex2_get_cols(z, ex, lp);
        prio_lo();
    }
    st.mark(4);
    dft8<false>(z);
    dft8<false>(z + 8);
    st.mark(5);
}

template <class TAB>
__device__ __forceinline__ void fft_inverse_1024(v2f* z, v2f* ex, int l, int a3, int g3, int kloB,
                                                 const TAB& tb, Stamps& st, const v2f* w2r = nullptr) {
    const LanePos lp(l, kloB);
    dft8<true>(z);
    dft8<true>(z + 8);
    st.mark(7);
    if (!(GL_ABL & 1)) {
        prio_hi();
        ex2_put_cols(z, ex, lp);
        ex2_get_rows(z, ex, lp);
        prio_lo();
    }
    st.mark(8);
    twiddle2<true>(z, tb, w2r);
    dft8<true>(z);
    dft8<true>(z + 8);
    st.mark(9);
    if (!(GL_ABL & 1)) {
        prio_hi();
        ex1_put_rows(z, ex, lp);
        ex1_get_cols(z, ex, lp);
        prio_lo();
    }
    st.mark(10);
    twiddle1<true>(z, tb);
    dft16_any<true>(z, tb);
    st.mark(11);
}

// rfft split + magnitude projection + inverse split on the 9 lane-local mirror pairs.
// lanes >= 1: P[c] = A[c], Q[c] = B[7-c]; lane 0 pairs inside its own two butterflies
// (oracle/sim_fast2048.py); unit 8 is lane 0's self-mirrored bin 512 (a dummy elsewhere).
// Lane 0 owns the two self-mirrored butterflies (k_lo = 0 and 64), so its nine units pair registers differently
// from every other lane's.  Instead of 24 two-instruction selects per frame, the twelve registers that differ
// are rotated for lane 0 alone: twelve v_mov_b64 under EXEC = lane 0, in an order that needs no temporary
// (the wave runs with all 64 lanes enabled: the kernels have no divergent control flow around this point).
#ifndef GL_L0_MOV
#define GL_L0_MOV 1
#endif
// d0 <- d1 <- ... <- d11 <- src   (for lane 0 only)
__device__ __forceinline__ void lane0_chain(v2f& d0, v2f& d1, v2f& d2, v2f& d3, v2f& d4, v2f& d5, v2f& d6, v2f& d7,
                                            v2f& d8, v2f& d9, v2f& d10, v2f& d11, v2f src) {
    asm("s_mov_b64 exec, 1\n\t"
        "v_mov_b64 %0, %1\n\tv_mov_b64 %1, %2\n\tv_mov_b64 %2, %3\n\tv_mov_b64 %3, %4\n\t"
        "v_mov_b64 %4, %5\n\tv_mov_b64 %5, %6\n\tv_mov_b64 %6, %7\n\tv_mov_b64 %7, %8\n\t"
        "v_mov_b64 %8, %9\n\tv_mov_b64 %9, %10\n\tv_mov_b64 %10, %11\n\tv_mov_b64 %11, %12\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(d8), "+v"(d9),
          "+v"(d10), "+v"(d11)
        : "v"(src));
}

// rfft split + magnitude projection + inverse split on the 9 lane-local mirror pairs.
// lanes >= 1: P[c] = A[c], Q[c] = B[7-c]; lane 0 pairs inside its own two butterflies
// (oracle/sim_fast2048.py); unit 8 is lane 0's self-mirrored bin 512 (a dummy elsewhere).
template <class TAB>
__device__ __forceinline__ void middle_project(v2f* z, bool l0, const float* m_lo, const float* m_hi,
                                               const TAB& tb) {
    v2f P[9], Q[9];
    const v2f z4 = z[4];
#if GL_L0_MOV
    // lane 0: P[4..7] = z[8..11], Q[0] = z[0], Q[1..3] = z[7..5], Q[4..7] = z[15..12]; with P[c] living in z[c]
    // and Q[c] in z[15-c] that is the rotation z4<-z8<-z12<-z5<-z9<-z13<-z6<-z10<-z14<-z7<-z11<-z15<-z0
    lane0_chain(z[4], z[8], z[12], z[5], z[9], z[13], z[6], z[10], z[14], z[7], z[11], z[15], z[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) { P[c] = z[c]; Q[c] = z[15 - c]; }
#else
    P[0] = z[0];
    Q[0] = sel(l0, z[0], z[15]);
#pragma unroll
    for (int c = 1; c < 4; ++c) { P[c] = z[c]; Q[c] = sel(l0, z[8 - c], z[15 - c]); }
#pragma unroll
    for (int c = 4; c < 8; ++c) { P[c] = sel(l0, z[8 + c - 4], z[c]); Q[c] = sel(l0, z[8 + 11 - c], z[15 - c]); }
#endif
    // w = -i*W2048^k.  2X[k] = (P + conj Q) + w*(P - conj Q), 2conj(X[M-k]) = the same with -w;
    // the projection is scale-free and commutes with conj; back: 2z[k] = (X+Y) + (X-Y)*conj(w),
    // 2z[M-k] = conj((X+Y) - (X-Y)*conj(w)).  The factor 2 is folded into the scaled window.
    auto unit = [&](int u, v2f w) {
        const v2f uu = pk_addc(P[u], Q[u]), vv = pk_subc(P[u], Q[u]);
        const v2f X = project(cmul_add(vv, w, uu), m_lo[u]);
        const v2f Y = project(cmul_sub(vv, w, uu), m_hi[u]);
        const v2f a2 = X + Y, b2 = X - Y;
        P[u] = cmulc_add(b2, w, a2);
        Q[u] = cmulc_subconj(b2, w, a2);
    };
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
        v2f wa, wb;
        tb.pair(kT2Twp + 2 * pp, wa, wb);
        unit(2 * pp, wa);
        unit(2 * pp + 1, wb);
    }
    // unit 8 = lane 0's bin 512, its own mirror (w = -1): the general formula collapses to 2 * project(z, m)
    P[8] = project(z4, m_lo[8] + m_lo[8]);
#if GL_L0_MOV
    // results live in R_c = P[c] (-> z[c]) and S_i = Q[7-i] (-> z[8+i]); lane 0 wants z4 = P8, z5..7 = Q3..1,
    // z8..11 = P4..7, z12..15 = Q7..4: the rotation S7<-S3<-R7<-S6<-S2<-R6<-S5<-S1<-R5<-S4<-S0<-R4<-P8
    lane0_chain(Q[0], Q[4], P[7], Q[1], Q[5], P[6], Q[2], Q[6], P[5], Q[3], Q[7], P[4], P[8]);
#pragma unroll
    for (int c = 0; c < 8; ++c) { z[c] = P[c]; z[8 + c] = Q[7 - c]; }
#else
    z[0] = P[0];
#pragma unroll
    for (int c = 1; c < 4; ++c) z[c] = P[c];
    z[4] = sel(l0, P[8], P[4]);
#pragma unroll
    for (int c = 5; c < 8; ++c) z[c] = sel(l0, Q[8 - c], P[c]);
#pragma unroll
    for (int i = 0; i < 4; ++i) z[8 + i] = sel(l0, P[4 + i], Q[7 - i]);
#pragma unroll
    for (int i = 4; i < 8; ++i) z[8 + i] = sel(l0, Q[11 - i], Q[7 - i]);
#endif
}

// Magnitudes as the register-ring kernel's lanes consume them.  Lane l of a frame needs the 17
// values m[k], m[1024-k] at its 8 mirror pairs k = l + 128u (lane 0: see middle_project) and
// m[512]: 17 dword loads per frame, rows 4100 bytes long so every load straddles three 128-byte
// lines.  The kernel is bound by the number of line requests a CU can keep in flight (TCP
// pending-stall 46 % of cycles), so gl_run first repacks the spectrogram once into
// [frame][quad q][lane][4] + [frame][1024 + lane]: four aligned dwordx4 loads and one dword per
// frame, 34 line requests instead of about 51.  4352 bytes per frame (6 % more than 4100).
constexpr int kMagPack = 1088;
// 1/M of the inverse transform and the factor 2 the split step leaves, folded into the magnitudes once
// (a power of two: exact), so the forward and the inverse window are the same table entries
constexpr float kMagScale = 1.0f / (2.0f * kM);
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) k_mag_repack(const float* __restrict__ mag, float* __restrict__ out, int T) {
    const int l = threadIdx.x;
    const long long fr = (long long)blockIdx.y * T + blockIdx.x;
    const float* mrow = mag + fr * (kM + 1);
    float* orow = out + fr * kMagPack;
    const int bA = l, bB = l ? l : -448;
    float v[16];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = (u < 4 ? bA : bB) + 128 * u;
        v[u] = mrow[k] * kMagScale;
        v[8 + u] = mrow[kM - k] * kMagScale;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<v4f*>(orow + (q * 64 + l) * 4) = (v4f){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    orow[1024 + l] = mrow[512] * kMagScale;
}

struct AlignedArgs {
    const float* x_in;      // [B][L]
    const float* tail_in;   // [B][nruns][N-H] or null (same run geometry as this launch)
    const float* magp;      // [B][T][kMagPack]: magnitudes repacked per lane (k_mag_repack)
    float* x_out;           // [B][L]
    float* tail_out;        // [B][nruns][N-H]
    const v2f* tab;         // [kT2Entries][64]
    int T, G, nruns, B;
    long long L;
};

#ifndef GL_MAG_PREFETCH
#define GL_MAG_PREFETCH 2
#endif
// 8 waves in one workgroup per CU = 2 waves per SIMD with up to 256 VGPRs each: the 47 per-lane constants of
// the frame loop then live in registers (238 VGPRs, no spills) instead of being re-read from LDS at every
// use, which removes a third of the LDS traffic and every table-read wait.  Measured on config 2:
// 12 waves + LDS table 0.149 ms, 8 waves + LDS table 0.154 ms, 8 waves + register table 0.1435 ms.
// GL_TAB_REG: 0 = constants read from LDS at every use, 1 = all 47 in registers, 2 = the FFT twiddles (22) in
// registers, windows and split twiddles from LDS -- what fits beside the second frame of the pipelined loop.
// GL_PIPE: 1 = software-pipelined loop over two frames (below).  Measured on config 2, 8 waves per CU:
// LDS table 0.154 ms; register table 0.1460; pipelined + LDS table 0.151; pipelined + twiddles in registers 0.1420.
#ifndef GL_TAB_REG
#define GL_TAB_REG 2
#endif
#ifndef GL_PIPE
#define GL_PIPE 1
#endif
#ifndef GL_RR_WAVES
#define GL_RR_WAVES 8
#endif
#ifndef GL_RR_OCC
#define GL_RR_OCC 1
#endif
constexpr int kRRWaves = GL_RR_WAVES;        // waves per workgroup; they only share the constant table

template <int HS>
__global__ void __launch_bounds__(64 * kRRWaves, GL_RR_OCC) k_gl2048_regring(AlignedArgs p) {
    // one LDS array (table + per-wave transpose scratch): 33 KB + 8 x 10.4 KB = 116 KB -> 1 WG / CU
    __shared__ __attribute__((aligned(16))) v2f smem[kT2Entries * 64 + kRRWaves * kRScratch];
    constexpr int H = 128 * HS;
    constexpr int OVR = 16 - HS;                 // rows shared with the next frame / next run
    constexpr int ovl = kN - H;
    const int l = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
    // runs of all utterances are numbered consecutively, so only the last workgroup has idle waves
    // The first global loads of the run (ring fill, magnitudes of its first frame) are issued BEFORE the
    // constant table is copied to LDS, so the two round trips to memory overlap; an idle wave of the last
    // workgroup goes through the motions on run 0 and leaves after the barrier.
    const int grun_raw = blockIdx.x * kRRWaves + wave;
    const bool active = grun_raw < p.B * p.nruns;
    const int grun = active ? grun_raw : 0;
    const int b = grun / p.nruns;
    const int run = grun - b * p.nruns;
    const int T = p.T;
    const int f0 = run * p.G;
    const int f1 = min(f0 + p.G, T);
    const int a3 = l & 7, g3 = l >> 3;
    const int kloB = l ? 128 - l : 64;
    const bool l0 = (l == 0);
    const int gh = p.G * H;
    const int nruns = p.nruns;

    const float* xr = p.x_in + (long long)b * p.L + 2 * l;
    const float* tr = p.tail_in ? p.tail_in + (long long)b * nruns * ovl + 2 * l : nullptr;

    // Row of 128 samples starting `rel` samples after this run's first sample.  The run that
    // produced it may have left part of its sum in a tail: the previous run's tail covers the
    // first N-H samples of this run's region, this run's own tail the first N-H of the next.
    const bool has_tail = (p.tail_in != nullptr);
    const int own_prev = (run >= 1) ? 1 : 0;
    const bool has_next = (run + 1 < nruns);
    const float* xrun = xr + (long long)f0 * H;
    const float* t_prev = tr ? tr + (long long)(run - own_prev) * ovl : nullptr;
    const float* t_own = tr ? tr + (long long)run * ovl : nullptr;
    // Branch-free in the data: the tail operand comes from the tail buffers or from a row of
    // zeros, chosen by wave-uniform pointer selects.  (Branches around the second load made the
    // compiler wait for the first one at the join: a load round trip per row per frame.)
    const float* zrow = reinterpret_cast<const float*>(p.tab + kT2Zero * 64) + 2 * l;
    auto tail_ptr = [&](int rel) -> const float* {
        const float* tp = zrow;
        if (has_tail) {
            const int rel_next = rel - gh;
            if (has_next && rel_next >= 0) {
                if (rel_next < ovl) tp = t_own + rel_next;
            } else if (own_prev && rel < ovl) {
                tp = t_prev + rel;
            }
        }
        return tp;
    };
    auto load_row = [&](int rel) -> v2f {
        return *reinterpret_cast<const v2f*>(xrun + rel) + *reinterpret_cast<const v2f*>(tail_ptr(rel));
    };

    v2f xraw[16], accr[16];
    {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            xraw[j] = load_row(128 * j);
            accr[j] = (v2f){0.0f, 0.0f};
        }
    }
    // magnitudes at the lane's 9 mirror pairs of frame `fr` (packed layout, see k_mag_repack)
    auto load_mags = [&](int fr, float* lo, float* hi) {
        const float* mrow = p.magp + ((long long)b * T + fr) * kMagPack + 4 * l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v4f t = (GL_ABL & 8) ? (v4f){1.0f, 2.0f, 3.0f, 4.0f} : *reinterpret_cast<const v4f*>(mrow + 256 * q);
            float* dst = (q < 2) ? lo + 4 * q : hi + 4 * (q - 2);
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        }
        lo[8] = (GL_ABL & 8) ? 1.0f : mrow[1024 - 3 * l];
        hi[8] = lo[8];
    };
    float m_lo[9], m_hi[9];
    load_mags(f0, m_lo, m_hi);
    for (int i = threadIdx.x; i < kT2Entries * 64; i += 64 * kRRWaves) smem[i] = p.tab[i];
    __syncthreads();
    if (!active) return;                         // wave-uniform; no barrier after this point
    v2f* ex = smem + kT2Entries * 64 + wave * kRScratch;
    const v2f* tabl = smem + 2 * l;
#if GL_TAB_REG == 1
    // 2 waves per SIMD leave 256 VGPRs per lane: the 47 constants the frame loop uses live in registers
    v2f treg[kT2A16F];
#pragma unroll
    for (int e = 0; e < kT2A16F; ++e) treg[e] = lds_rd(&tabl[TE(e)]);
    const TabReg tb{treg};
#elif GL_TAB_REG == 2
    v2f treg[kT2Twp - kT2Tw1];
#pragma unroll
    for (int e = kT2Tw1; e < kT2Twp; ++e) treg[e - kT2Tw1] = lds_rd(&tabl[TE(e)]);
    const TabMixed tb{tabl, treg};
#else
    const TabLds tb{tabl};
#endif
#if GL_TW2_REG
    v2f w2reg[7];
#pragma unroll
    for (int d = 1; d < 8; ++d) w2reg[d - 1] = tb.get(kT2Tw2 + d - 1);
    const v2f* w2r = w2reg;
#else
    const v2f* w2r = nullptr;
#endif

    // Drain the ring-fill loads here.  Otherwise the loop header inherits "xraw[8..15] still in
    // flight" from this prologue and the compiler guards every frame's window multiply with
    // s_waitcnt vmcnt(24..17), which in steady state waits for the rows it has just requested
    // for the NEXT frame: one exposed global-load round trip per frame.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
    Stamps st;
    st.start();

#if GL_PIPE
    // Software pipeline over two frames.  Frames of one iteration are independent (every frame reads x_in
    // and adds into x_out), so the forward transform of frame n+1 ("A") is interleaved with the projection
    // and inverse transform of frame n ("B"): each of the four LDS transposes is issued and then followed
    // by arithmetic of the OTHER frame, so its round trip is covered by work of the same wave instead of
    // depending on what the one other wave of the SIMD happens to be doing.  Both frames use the same
    // scratch: LDS operations of a wave execute in program order, and the order below alternates
    // A.ex1, B.ex2, A.ex2, B.ex1, each written and read completely before the next is written.
    const LanePos lp(l, kloB);
    auto window_in = [&](v2f* z) {
#pragma unroll
        for (int j = 0; j < 16; ++j) z[j] = xraw[j] * tb.get(kT2Win + j);
    };
    auto rows_request = [&](int frame, bool live, v2f* xn, v2f* xt) {      // the HS rows frame `frame` adds
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) {
            const int rel = (frame - 1 - f0) * H + kN + 128 * jj;
            const float* xp = live ? xrun + rel : zrow;
            const float* tp = live ? tail_ptr(rel) : zrow;
            xn[jj] = *reinterpret_cast<const v2f*>(xp);
            xt[jj] = *reinterpret_cast<const v2f*>(tp);
        }
    };
    auto ring_advance = [&](const v2f* xn, const v2f* xt) {
#pragma unroll
        for (int j = 0; j < OVR; ++j) xraw[j] = xraw[j + HS];
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) xraw[OVR + jj] = xn[jj] + xt[jj];
    };
    auto emit = [&](v2f* z, int n) {                 // window, overlap-add, store the HS finished rows
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v2f w = tb.get(kT2Win + j);              // 1/2048 is folded into the packed magnitudes
            accr[j].x = fmaf(z[j].x, w.x, accr[j].x);
            accr[j].y = fmaf(z[j].y, w.y, accr[j].y);
        }
        float* orow = p.x_out + (long long)b * p.L + (long long)n * H + 2 * l;
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) *reinterpret_cast<v2f*>(orow + 128 * jj) = accr[jj];
#pragma unroll
        for (int j = 0; j < OVR; ++j) accr[j] = accr[j + HS];
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) accr[OVR + jj] = (v2f){0.0f, 0.0f};
    };

    // prologue: forward transform of the run's first frame
    v2f zf[16];
    {
        v2f xn[HS], xt[HS];
        rows_request(f0 + 1, f0 + 1 < f1, xn, xt);
        window_in(zf);
        fft_forward_1024(zf, ex, l, a3, g3, kloB, tb, st, w2r);
        ring_advance(xn, xt);                        // xraw = input of frame f0+1
    }
    for (int n = f0; n + 1 < f1; ++n) {
        v2f xn[HS], xt[HS];
        rows_request(n + 2, n + 2 < f1, xn, xt);
        v2f zA[16];
        window_in(zA);                               // A: frame n+1
        dft16_any<false>(zA, tb);
        twiddle1<false>(zA, tb);
        ex1_put_cols(zA, ex, lp);
        ex1_get_rows(zA, ex, lp);
        middle_project(zf, l0, m_lo, m_hi, tb);      // B: frame n
        load_mags(n + 1, m_lo, m_hi);
        dft8<true>(zf);
        dft8<true>(zf + 8);
        ex2_put_cols(zf, ex, lp);
        ex2_get_rows(zf, ex, lp);
        dft8<false>(zA);                             // A
        dft8<false>(zA + 8);
        twiddle2<false>(zA, tb, w2r);
        ex2_put_rows(zA, ex, lp);
        ex2_get_cols(zA, ex, lp);
        twiddle2<true>(zf, tb, w2r);                 // B
        dft8<true>(zf);
        dft8<true>(zf + 8);
        ex1_put_rows(zf, ex, lp);
        ex1_get_cols(zf, ex, lp);
        dft8<false>(zA);                             // A: spectrum of frame n+1 complete
        dft8<false>(zA + 8);
        twiddle1<true>(zf, tb);                      // B
        dft16_any<true>(zf, tb);
        emit(zf, n);
        ring_advance(xn, xt);                        // xraw = input of frame n+2
#pragma unroll
        for (int j = 0; j < 16; ++j) zf[j] = zA[j];
    }
    {   // epilogue: the run's last frame has no successor to overlap with
        middle_project(zf, l0, m_lo, m_hi, tb);
        fft_inverse_1024(zf, ex, l, a3, g3, kloB, tb, st, w2r);
        emit(zf, f1 - 1);
    }
#else
    for (int n = f0; n < f1; ++n) {
        const long long sn = (long long)n * H;

        // prefetch the HS rows the next frame adds
        const bool more = (n + 1 < f1);
        st.mark(15);
        // (both halves of each row are requested here and summed only at the end of the frame;
        // past the run's last frame they come from the zero row: no branch around the loads)
        v2f xnew[HS], xnew_t[HS];
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) {
            const int rel = (n - f0) * H + kN + 128 * jj;
            const float* xp = more ? xrun + rel : zrow;
            const float* tp = more ? tail_ptr(rel) : zrow;
            xnew[jj] = *reinterpret_cast<const v2f*>(xp);
            xnew_t[jj] = *reinterpret_cast<const v2f*>(tp);
        }
        // request the NEXT frame's magnitudes (this frame's were requested a frame ago: an HBM
        // round trip is longer than the forward transform that used to cover it)
        st.mark(13);
#if GL_MAG_PREFETCH == 1   // variant: request at the frame top into a second register set
        float mn_lo[9], mn_hi[9];
        load_mags(more ? n + 1 : n, mn_lo, mn_hi);
#endif

        v2f z[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) z[j] = xraw[j] * tb.get(kT2Win + j);
        st.mark(0);
        fft_forward_1024(z, ex, l, a3, g3, kloB, tb, st, w2r);
        if (!(GL_ABL & 2)) middle_project(z, l0, m_lo, m_hi, tb);
        // the magnitude registers are free again: request the next frame's now, so the loads have
        // the inverse and the next forward transform to land (an HBM round trip under load is
        // longer than one transform) at no cost in registers
#if GL_MAG_PREFETCH == 2
        load_mags(more ? n + 1 : n, m_lo, m_hi);
#endif
        st.mark(6);
        fft_inverse_1024(z, ex, l, a3, g3, kloB, tb, st, w2r);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v2f w = tb.get(kT2Win + j);      // 1/2048 is folded into the packed magnitudes
            accr[j].x = fmaf(z[j].x, w.x, accr[j].x);
            accr[j].y = fmaf(z[j].y, w.y, accr[j].y);
        }
        // rows 0..HS-1 are complete: every sample is written exactly once
        float* orow = p.x_out + (long long)b * p.L + sn + 2 * l;
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) *reinterpret_cast<v2f*>(orow + 128 * jj) = accr[jj];
#pragma unroll
        for (int j = 0; j < OVR; ++j) { accr[j] = accr[j + HS]; xraw[j] = xraw[j + HS]; }
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) { accr[OVR + jj] = (v2f){0.0f, 0.0f}; xraw[OVR + jj] = xnew[jj] + xnew_t[jj]; }
#if GL_MAG_PREFETCH == 1
#pragma unroll
        for (int u = 0; u < 9; ++u) { m_lo[u] = mn_lo[u]; m_hi[u] = mn_hi[u]; }
#endif
        st.mark(12);
    }

#endif

    // partial sums of the N-H samples after the run's last frame
    const long long se = (long long)f1 * H;
    st.flush(l);
    if (f1 == T) {
        float* orow = p.x_out + (long long)b * p.L + se + 2 * l;
#pragma unroll
        for (int j = 0; j < OVR; ++j) *reinterpret_cast<v2f*>(orow + 128 * j) = accr[j];
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) *reinterpret_cast<v2f*>(orow + 128 * (OVR + jj)) = (v2f){0.0f, 0.0f};
    } else {
        float* trow = p.tail_out + ((long long)b * nruns + run) * ovl + 2 * l;
#pragma unroll
        for (int j = 0; j < OVR; ++j) *reinterpret_cast<v2f*>(trow + 128 * j) = accr[j];
    }
}


// =====================================================================================
// Small problems at n_fft 2048, hop 128*HS -- k_gl2048_frames<HS> + k_ola_frames.  Same idea as k_gl1024_frames
// below: when B*T frames cannot fill the GPU with runs (one utterance; one rank's chunk of a long utterance), a
// wave owns ONE frame, the state between iterations is the windowed inverse frames y[b][t][2048], and a wave
// rebuilds its input frame from the previous iteration's y while loading it.  With the hop a multiple of the
// 128-sample register row everything is row-granular and wave-uniform: row j of frame n is global row
// R = n HS + j, covered by frames m = ceil((R-15)/HS) .. floor(R/HS) at their row R - m HS (ascending m, the order
// of the reference's += loop) -- 16/HS terms per row, no per-lane conditions, one buffer load each with a
// scalar offset.  Frames past either end of the utterance contribute zero rows.  Hop 128 (256 loads per lane) is
// left to the run kernel.
struct Frames2048Args {
    const float* x_in;      // [B][L] start signal (first iteration) or null
    const float* y_in;      // [B][T][2048] windowed inverse frames of the previous iteration (x_in == null)
    const float* magp;      // [B][T][kMagPack] packed magnitudes (k_mag_repack)
    float* y_out;           // [B][T][2048]
    const v2f* tab;
    int T, B;
    long long L;
};
template <int HS>
__global__ void __launch_bounds__(64 * kRRWaves) k_gl2048_frames(Frames2048Args p) {
    __shared__ __attribute__((aligned(16))) v2f smem[kT2Entries * 64 + kRRWaves * kRScratch];
    constexpr int H = 128 * HS;
    const int l = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = p.T;
    const long long gf_raw = (long long)blockIdx.x * kRRWaves + wave;
    const bool active = gf_raw < (long long)p.B * T;
    const long long gf = active ? gf_raw : 0;    // an idle wave of the last workgroup goes through the motions on frame 0
    const int b = (int)(gf / T);
    const int n = (int)(gf - (long long)b * T);

    // operands first (one batch of loads), then the constant table, so the round trips overlap
    float m_lo[9], m_hi[9];
    {
        const float* mrow = p.magp + ((long long)b * T + n) * kMagPack + 4 * l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4f t = *reinterpret_cast<const v4f*>(mrow + 256 * q);
            float* dst = (q < 2) ? m_lo + 4 * q : m_hi + 4 * (q - 2);
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        }
        m_lo[8] = mrow[1024 - 3 * l];
        m_hi[8] = m_lo[8];
    }
    v2f z[16];
    if (p.x_in != nullptr) {
        const float* xr = p.x_in + (long long)b * p.L + (long long)n * H + 2 * l;
#pragma unroll
        for (int j = 0; j < 16; ++j) z[j] = *reinterpret_cast<const v2f*>(xr + 128 * j);
    } else {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.y_in + (long long)b * T * kN), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            z[j] = (v2f){0.0f, 0.0f};
            // d = m - n from -floor((15 - j) / HS) to floor(j / HS), compile-time per j
#pragma unroll
            for (int d = -((15 - j) / HS); d <= j / HS; ++d) {
                const int m = n + d;
                const bool ok = m >= 0 && m < T;                      // wave-uniform
                const int mc = ok ? m : n;
                const int row = j - d * HS;                           // 0..15
                // byte offset of (frame mc, row): below 2^31 for B*T*8 KB per utterance up to 2 GB
                const int soff = (mc * kN + row * 128) * 4;
                const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs, 8 * l, soff, 0));
                z[j] += ok ? v : (v2f){0.0f, 0.0f};
            }
        }
    }

    for (int i = threadIdx.x; i < kT2Entries * 64; i += 64 * kRRWaves) smem[i] = p.tab[i];
    __syncthreads();
    if (!active) return;                         // wave-uniform; no barrier after this point
    v2f* ex = smem + kT2Entries * 64 + wave * kRScratch;
    const v2f* tabl = smem + 2 * l;
    v2f treg[kT2Twp - kT2Tw1];
#pragma unroll
    for (int e = kT2Tw1; e < kT2Twp; ++e) treg[e - kT2Tw1] = lds_rd(&tabl[TE(e)]);
    const TabMixed tb{tabl, treg};
    const int a3 = l & 7, g3 = l >> 3;
    const int kloB = l ? 128 - l : 64;
    const bool l0 = (l == 0);

#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = z[j] * tb.get(kT2Win + j);
    Stamps st;
    fft_forward_1024(z, ex, l, a3, g3, kloB, tb, st);
    middle_project(z, l0, m_lo, m_hi, tb);
    fft_inverse_1024(z, ex, l, a3, g3, kloB, tb, st);
    float* yo = p.y_out + ((long long)b * T + n) * kN + 2 * l;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const v2f w = tb.get(kT2Win + j);                             // 1/2048 is folded into the packed magnitudes
        *reinterpret_cast<v2f*>(yo + 128 * j) = z[j] * w;
    }
}

// x[b][s] = sum over frames m (ascending) of y[b][m][s - mH]: the overlap-add of audio_utilities.py:111-113 as a
// gather; the last H samples no frame covers come out zero, like the reference's.
__global__ void __launch_bounds__(256) k_ola_frames(const float* __restrict__ y, float* __restrict__ x, long long L, int N,
                                                    int H, int T) {
    const int b = blockIdx.y;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    const float* yb = y + (long long)b * T * N;
    long long m_lo = (s - N + H) / H;             // ceil((s - N + 1) / H) for s - N + 1 > 0
    if (s - N + 1 <= 0) m_lo = 0;
    long long m_hi = s / H;
    if (m_hi > T - 1) m_hi = T - 1;
    float v = 0.0f;
    for (long long m = m_lo; m <= m_hi; ++m) v += yb[m * N + (s - m * H)];
    x[(long long)b * L + s] = v;
}

// =====================================================================================
// n_fft = 1024, any hop (the reference's params.py defaults are 1024 / 275).
//
// Two consecutive REAL frames ride one 1024-point COMPLEX FFT: z = frame_a + i*frame_b.
// With S = Z[k] + conj(Z[M-k]) = 2*Xa[k] and D = Z[k] - conj(Z[M-k]) = 2i*Xb[k] the
// projection of both frames is  Z'[k] = S*sa + D*sb,  Z'[M-k] = conj(S*sa - D*sb)
// (sa = mag_a/|S|, sb = mag_b/|D|, magnitude index min(k, M-k)); no twiddle is needed
// and the mirror pairs are lane-local in the same lane map as the 2048 kernels, so the
// FFT core is shared (oracle/sim_fast2048.py: check_two_frames).  The real part of the
// inverse transform is frame a's output, the imaginary part frame b's.
// x and the overlap-add accumulator live in per-wave LDS rings (hop 275 is not a
// multiple of the 64-sample register rows).
// =====================================================================================
constexpr int kN1 = 1024;
constexpr int kP1Waves = 4;
constexpr int kP1Ring = 2048;      // floats; >= N + H for every hop <= N
constexpr int kP1WaveLds = kRScratch + kP1Ring;   // v2f units per wave: transpose scratch + two float rings

struct Pair1024Args {
    const float* x_in;      // [B][L]
    const float* tail_in;   // [B][nruns][N-H] or null (same run geometry as this launch)
    const float* mag;       // [B][T][513]
    float* x_out;           // [B][L]
    float* tail_out;        // [B][nruns][N-H]
    const v2f* tab;         // [kT2Entries][64]; window entries hold (w[l+64j], w[l+64j])
    int H, T, G, nruns;
    long long L;
};

// projection step for two frames sharing one complex spectrum (see header above), in packed form:
//   S = P + conj Q = 2 Xa,  D = P - conj Q = 2i Xb;  Ya = project(S, ma),  Yb = i project(-i D, mb) = project'(D, mb)
//   where project' nudges the IMAGINARY part, so that a zero bin of frame b becomes i (mb, 0) as before;
//   P' = Ya + Yb,  Q' = conj(Ya - Yb).  ka / kb = 0 force the spectrum of an all-zero frame to exactly zero.
// Lane 0's self-mirrored pairs are rotated into place by two masked move chains (see middle_project).
__device__ __forceinline__ v2f project_im(v2f X, float m) {
    const v2f Xe = X + (v2f){0.0f, 0x1p-60f};
    const float r2 = fmaxf(fmaf(Xe.x, Xe.x, Xe.y * Xe.y), 0x1p-124f);
    const float sc = m * __builtin_amdgcn_rsqf(r2);
    return Xe * (v2f){sc, sc};
}
__device__ __forceinline__ void middle_two_frames(v2f* z, bool l0, const float* ma, const float* mb, float ka,
                                                  float kb) {
    v2f P[9], Q[9];
    const v2f z4 = z[4];
    lane0_chain(z[4], z[8], z[12], z[5], z[9], z[13], z[6], z[10], z[14], z[7], z[11], z[15], z[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) { P[c] = z[c]; Q[c] = z[15 - c]; }
    P[8] = z4;
    Q[8] = z4;
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        const v2f S = pk_addc(P[u], Q[u]) * ka;
        const v2f D = pk_subc(P[u], Q[u]) * kb;
        const v2f Ya = project(S, ma[u]);
        const v2f Yb = project_im(D, mb[u]);
        P[u] = Ya + Yb;
        Q[u] = pk_subcc(Ya, Yb);
    }
    lane0_chain(Q[0], Q[4], P[7], Q[1], Q[5], P[6], Q[2], Q[6], P[5], Q[3], Q[7], P[4], P[8]);
#pragma unroll
    for (int c = 0; c < 8; ++c) { z[c] = P[c]; z[8 + c] = Q[7 - c]; }
    (void)l0;
}

__global__ void __launch_bounds__(64 * kP1Waves, 1) k_gl1024_pair(Pair1024Args p) {
    // table 41 KB + 4 x (10.4 KB transpose scratch + 8 KB x ring + 8 KB accumulator ring) = 147 KB
    __shared__ __attribute__((aligned(16))) v2f smem[kT2EntriesPair * 64 + kP1Waves * kP1WaveLds];
    const int l = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int run = blockIdx.x * kP1Waves + wave;
    for (int i = threadIdx.x; i < kT2EntriesPair * 64; i += 64 * kP1Waves) smem[i] = p.tab[i];
    __syncthreads();
    if (run >= p.nruns) return;                  // wave-uniform; no barrier after this point
    v2f* ex = smem + kT2EntriesPair * 64 + wave * kP1WaveLds;
    float* xring = reinterpret_cast<float*>(ex + kRScratch);
    float* acc = xring + kP1Ring;
    const v2f* tabl = smem + 2 * l;

    const int H = p.H, T = p.T;
    const int f0 = run * p.G;
    const int f1 = min(f0 + p.G, T);
    const int ovl = kN1 - H;
    const int a3 = l & 7, g3 = l >> 3;
    const int kloB = l ? 128 - l : 64;
    const bool l0 = (l == 0);
    const int g_h = p.G * H;
    const float* xr = p.x_in + (long long)b * p.L;
    const float* tr = p.tail_in ? p.tail_in + (long long)b * p.nruns * ovl : nullptr;

#pragma unroll
    for (int i = 0; i < kP1Ring / 64; ++i) acc[l + 64 * i] = 0.0f;

    // magnitude index of the lane's 9 units: min(k, 1024-k)
    int kk[9];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        int k = l ? (l + 128 * u) : (u < 4 ? 128 * u : 64 + 128 * (u - 4));
        kk[u] = k <= 512 ? k : 1024 - k;
    }
    kk[8] = 512;

    long long filled = (long long)f0 * H;        // x ring holds [.., filled)
    for (int n = f0; n < f1; n += 2) {
        const bool two = (n + 1 < f1);
        const long long sn = (long long)n * H;
        const long long need = sn + kN1 + (two ? H : 0);
        for (long long s = filled + l; s < need; s += 64)
            xring[(int)(s & (kP1Ring - 1))] = fetch_x(xr, tr, s, g_h, p.nruns, ovl);
        filled = need;

        float ma[9], mb[9];
        {
            const float* ra = p.mag + ((long long)b * T + n) * (kN1 / 2 + 1);
            const float* rb = ra + (two ? (kN1 / 2 + 1) : 0);
#pragma unroll
            for (int u = 0; u < 9; ++u) { ma[u] = ra[kk[u]]; mb[u] = two ? rb[kk[u]] : 0.0f; }
        }
        __builtin_amdgcn_wave_barrier();

        v2f z[16];
        bool a_nz = false, b_nz = false;
        {
            const int base_a = (int)(sn & (kP1Ring - 1));
            const int base_b = (int)((sn + H) & (kP1Ring - 1));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float xa = xring[(base_a + l + 64 * j) & (kP1Ring - 1)];
                float xb = two ? xring[(base_b + l + 64 * j) & (kP1Ring - 1)] : 0.0f;
                a_nz |= (xa != 0.0f);
                b_nz |= (xb != 0.0f);
                v2f w = tabl[TE(kT2Win + j)];
                z[j].x = xa * w.x;
                z[j].y = xb * w.y;
            }
        }
        // an all-zero input frame must give an exactly zero spectrum (phase 0 after projection,
        // audio_utilities.py:181): its partner's rounding noise must not set its phase
        const float ka = __any(a_nz) ? 1.0f : 0.0f;
        const float kb = __any(b_nz) ? 1.0f : 0.0f;

        Stamps st;
        fft_forward_1024(z, ex, l, a3, g3, kloB, TabLds{tabl}, st);
        middle_two_frames(z, l0, ma, mb, ka, kb);
        fft_inverse_1024(z, ex, l, a3, g3, kloB, TabLds{tabl}, st);

        // window, 1/1024, overlap-add: frame a then frame b (frame order, like the reference)
        {
            const int base_a = (int)(sn & (kP1Ring - 1));
            const int base_b = (int)((sn + H) & (kP1Ring - 1));
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v2f w = tabl[TE(kT2WinSc + j)];
                acc[(base_a + l + 64 * j) & (kP1Ring - 1)] += z[j].x * w.x;
            }
            __builtin_amdgcn_wave_barrier();
            if (two) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    v2f w = tabl[TE(kT2WinSc + j)];
                    acc[(base_b + l + 64 * j) & (kP1Ring - 1)] += z[j].y * w.y;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // samples [sn, sn + H) (or 2H) are complete: emit once and clear
            const int done = two ? 2 * H : H;
            float* orow = p.x_out + (long long)b * p.L;
            for (int i = l; i < done; i += 64) {
                int q = (int)((sn + i) & (kP1Ring - 1));
                orow[sn + i] = acc[q];
                acc[q] = 0.0f;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // partial sums of the N-H samples after the run's last frame
    const long long se = (long long)f1 * H;
    if (f1 == T) {
        float* orow = p.x_out + (long long)b * p.L;
        for (int i = l; i < ovl; i += 64) orow[se + i] = acc[(int)((se + i) & (kP1Ring - 1))];
        for (int i = l; i < H; i += 64) orow[se + ovl + i] = 0.0f;
    } else {
        float* trow = p.tail_out + ((long long)b * p.nruns + run) * ovl;
        for (int i = l; i < ovl; i += 64) trow[i] = acc[(int)((se + i) & (kP1Ring - 1))];
    }
}

// Second version of the n_fft-1024 kernel: 8 waves per CU instead of 4.  The x ring is gone (a frame's samples
// come straight from global memory: consecutive frames overlap, so they hit L1/L2), the FFT twiddles live in
// registers and only the 16 window entries are copied to LDS, and 1/1024 is folded into the magnitudes so the
// inverse window is the forward window: 8 KB + 8 x (10.4 KB scratch + 8 KB accumulator ring) = 155 KB.
// Runs of all utterances are numbered consecutively, like the register-ring kernel's.
constexpr int kP2Waves = 8;
constexpr int kP2WaveLds = kRScratch + kP1Ring / 2;      // v2f units: transpose scratch + one float ring
struct Pair2Args {
    Pair1024Args a;
    int B;
    float* x_zero;          // [B][L] the buffer the next launch writes: its seam regions are cleared here (or null)
};
__global__ void __launch_bounds__(64 * kP2Waves, 1) k_gl1024_pair2(Pair2Args pp) {
    const Pair1024Args& p = pp.a;
    __shared__ __attribute__((aligned(16))) v2f smem[16 * 64 + kP2Waves * kP2WaveLds];
    const int l = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grun = blockIdx.x * kP2Waves + wave;
    for (int i = threadIdx.x; i < 16 * 64; i += 64 * kP2Waves) smem[i] = p.tab[i];      // window entries 0..15
    __syncthreads();
    if (grun >= pp.B * p.nruns) return;          // wave-uniform; no barrier after this point
    const int b = grun / p.nruns;
    const int run = grun - b * p.nruns;
    v2f* ex = smem + 16 * 64 + wave * kP2WaveLds;
    float* acc = reinterpret_cast<float*>(ex + kRScratch);
    const v2f* tabl = smem + 2 * l;
    v2f treg[kT2Twp - kT2Tw1];
    {
        const v2f* tg = p.tab + 2 * l;
#pragma unroll
        for (int e = kT2Tw1; e < kT2Twp; ++e) treg[e - kT2Tw1] = tg[TE(e)];
    }
    const TabMixed tb{tabl, treg};
    // both frames of a pair see the same window value, so the 16 per lane are 16 scalars: kept in registers
    // (48 LDS reads per pair otherwise)
    float wreg[16];
    {
        const v2f* tg = p.tab + 2 * l;
#pragma unroll
        for (int j = 0; j < 16; ++j) wreg[j] = tg[TE(kT2Win + j)].x;
    }

    const int H = p.H, T = p.T;
    const int f0 = run * p.G;
    const int f1 = min(f0 + p.G, T);
    const int ovl = kN1 - H;
    const int a3 = l & 7, g3 = l >> 3;
    const int kloB = l ? 128 - l : 64;
    const bool l0 = (l == 0);
    const float* xr = p.x_in + (long long)b * p.L;
    // x_in is complete.  Run seams: the N-H samples after a run's last frame belong to the next run's region; both
    // runs ADD their partial sums there with hardware float atomics onto zeros (two terms onto 0: the result does
    // not depend on which arrives first), and every launch clears the seam regions of the buffer the NEXT launch
    // will write (x_zero; three signal buffers rotate).  Earlier forms: tails folded by the consumer (96 loads
    // with per-lane conditions in front of every pair at a seam, issued three at a time between full waits -- 32
    // trips to memory in a row, half of all pairs at config 5), then folded by a separate in-place kernel between
    // iterations (5.9 us of launch per iteration for 1.5 M samples).
    const long long s_lo = (long long)f0 * H;                 // first sample of this run's region
    float* xo = p.x_out + (long long)b * p.L;
    const bool seamed = run >= 1 && ovl > 0;                  // [s_lo, s_lo + ovl) also receives the previous run's sums
    auto put = [&](long long s, float v, bool on) {           // sample s of the output is complete (for this run)
        if (!on) return;
        if (seamed && s < s_lo + ovl) unsafeAtomicAdd(xo + s, v);
        else xo[s] = v;
    };
    if (pp.x_zero != nullptr && seamed) {
        float* zr = pp.x_zero + (long long)b * p.L + s_lo;
        for (int i = l; i < ovl; i += 64) zr[i] = 0.0f;
    }
    auto ldb = [](__amdgpu_buffer_rsrc_t rs, int byte_off) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
    };

#pragma unroll
    for (int i = 0; i < kP1Ring / 64; ++i) acc[l + 64 * i] = 0.0f;

    int kk[9];                                   // magnitude index of the lane's 9 units: min(k, 1024-k)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        int k = l ? (l + 128 * u) : (u < 4 ? 128 * u : 64 + 128 * (u - 4));
        kk[u] = k <= 512 ? k : 1024 - k;
    }
    kk[8] = 512;

    // Operands of pair n (raw samples of both frames, magnitudes): requested one pair ahead, so their round trip
    // to memory runs under the transforms of the pair before.
    auto fetch = [&](int n, v2f (&xz)[16], float (&fa)[9], float (&fb)[9]) {
        const bool two = (n + 1 < f1);
        const long long sn = (long long)n * H;
        const float* ra = p.mag + ((long long)b * T + n) * (kN1 / 2 + 1);
        const float* rb = ra + (two ? (kN1 / 2 + 1) : 0);
#pragma unroll
        for (int u = 0; u < 9; ++u) { fa[u] = ra[kk[u]]; fb[u] = rb[kk[u]]; }
        const long long sb = two ? sn + H : sn;                       // a lone last frame: b re-reads a (discarded)
        // buffer loads: wave-uniform base in scalar registers + one 32-bit lane offset for all 32
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(xr + sn), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(xr + sb), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 16; ++j) { xz[j].x = ldb(rs_a, 4 * (l + 64 * j)); xz[j].y = ldb(rs_b, 4 * (l + 64 * j)); }
    };
    v2f xn[16];
    float fa[9], fb[9];
    if (f0 < f1) fetch(f0, xn, fa, fb);

    for (int n = f0; n < f1; n += 2) {
        const bool two = (n + 1 < f1);
        const long long sn = (long long)n * H;
        float ma[9], mb[9];
        v2f z[16];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            ma[u] = fa[u] * (1.0f / kM);                              // 1/M of the inverse transform (exact)
            mb[u] = two ? fb[u] * (1.0f / kM) : 0.0f;
        }
        bool a_nz = false, b_nz = false;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            z[j].x = xn[j].x;
            z[j].y = two ? xn[j].y : 0.0f;
            a_nz |= (z[j].x != 0.0f);
            b_nz |= (z[j].y != 0.0f);
            z[j] = z[j] * wreg[j];
        }
        if (n + 2 < f1) fetch(n + 2, xn, fa, fb);                     // wave-uniform
        // an all-zero input frame must give an exactly zero spectrum (phase 0 after projection,
        // audio_utilities.py:181): its partner's rounding noise must not set its phase
        const float ka = __any(a_nz) ? 1.0f : 0.0f;
        const float kb = __any(b_nz) ? 1.0f : 0.0f;

        Stamps st;
        fft_forward_1024(z, ex, l, a3, g3, kloB, tb, st);
        if (!(GL_ABL & 2)) middle_two_frames(z, l0, ma, mb, ka, kb);
        fft_inverse_1024(z, ex, l, a3, g3, kloB, tb, st);

        // window, overlap-add: frame a then frame b (frame order, like the reference).  A sample is complete once
        // the last frame covering it has been added: the first H samples of frame a after a (frame b starts H
        // later), the first H of frame b after b (the next pair starts 2H later).  Those leave for global memory
        // straight from the add and their ring slots are cleared in the same pass -- no separate emit loop.
        if (!(GL_ABL & 16)) {
            const int base_a = (int)(sn & (kP1Ring - 1));
            const int base_b = (int)((sn + H) & (kP1Ring - 1));
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int q = (base_a + l + 64 * j) & (kP1Ring - 1);
                const float v = acc[q] + z[j].x * wreg[j];
                const bool fin = l + 64 * j < H;
                acc[q] = fin ? 0.0f : v;
                put(sn + l + 64 * j, v, fin);
            }
            __builtin_amdgcn_wave_barrier();
            if (two) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int q = (base_b + l + 64 * j) & (kP1Ring - 1);
                    const float v = acc[q] + z[j].y * wreg[j];
                    const bool fin = l + 64 * j < H;
                    acc[q] = fin ? 0.0f : v;
                    put(sn + H + l + 64 * j, v, fin);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // partial sums of the N-H samples after the run's last frame: the end of the signal, or the next run's seam
    const long long se = (long long)f1 * H;
    if (f1 == T) {
        for (int i = l; i < ovl; i += 64) put(se + i, acc[(int)((se + i) & (kP1Ring - 1))], true);
        for (int i = l; i < H; i += 64) xo[se + ovl + i] = 0.0f;
    } else {
        for (int i = l; i < ovl; i += 64) unsafeAtomicAdd(xo + se + i, acc[(int)((se + i) & (kP1Ring - 1))]);
    }
}

// Low-latency form of the n_fft-1024 loop for problems too small to fill the GPU with runs of frames (config 1:
// one utterance, 641 frames -> 160 runs of 4 frames, 24.5 us per iteration however few CUs are busy).  Here a
// wave owns ONE pair of frames and nothing is carried from frame to frame: the state between iterations is not
// the signal but the windowed inverse frames y[b][t][1024], and a wave rebuilds its two input frames from the
// previous iteration's y as it loads them,
//     x[nH + i] = sum over frames m (ascending, the order of the reference's += loop) of y[m][(n - m)H + i],
// (up to ceil(N/H) terms per sample, all L2 hits), so every pair of an iteration runs at once and there are no
// run seams, tails or rings.  The first iteration reads the start signal instead; k_ola_frames folds the last y
// into the output signal.  Results do not depend on any work split.
constexpr int kF1Waves = 4;
constexpr int kF1MaxReach = 8;               // frames a sample can reach back/forward: floor((N-1)/H) <= 8, H >= 114
struct Frames1024Args {
    const float* x_in;      // [B][L] start signal (first iteration) or null
    const float* y_in;      // [B][T][1024] windowed inverse frames of the previous iteration (x_in == null)
    const float* mag;       // [B][T][513]
    float* y_out;           // [B][T][1024]
    const v2f* tab;
    int H, T, B, reach;     // reach = (N-1)/H
    long long L;
};
// REACH = floor((N-1)/H), a template parameter so that the gather is one straight block of (REACH+1) x 32 loads
// per lane, all in flight together: y was written by CUs of other XCDs, so every dependent batch of loads costs
// a trip to memory (about 2 us), and with a run-time loop there were REACH+1 of them in a row.
template <int REACH>
__global__ void __launch_bounds__(64 * kF1Waves) k_gl1024_frames(Frames1024Args p) {
    __shared__ __attribute__((aligned(16))) v2f smem[16 * 64 + kF1Waves * kRScratch];
    const int l = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16 * 64; i += 64 * kF1Waves) smem[i] = p.tab[i];      // window entries 0..15
    __syncthreads();
    const int T = p.T, H = p.H;
    const int npairs = (T + 1) >> 1;
    const long long gp = (long long)blockIdx.x * kF1Waves + wave;
    if (gp >= (long long)p.B * npairs) return;   // wave-uniform; no barrier after this point
    const int b = (int)(gp / npairs);
    const int n = 2 * (int)(gp - (long long)b * npairs);
    const bool two = (n + 1 < T);
    v2f* ex = smem + 16 * 64 + wave * kRScratch;
    const v2f* tabl = smem + 2 * l;
    v2f treg[kT2Twp - kT2Tw1];
    {
        const v2f* tg = p.tab + 2 * l;
#pragma unroll
        for (int e = kT2Tw1; e < kT2Twp; ++e) treg[e - kT2Tw1] = tg[TE(e)];
    }
    const TabMixed tb{tabl, treg};
    const int a3 = l & 7, g3 = l >> 3;
    const int kloB = l ? 128 - l : 64;
    const bool l0 = (l == 0);

    float ma[9], mb[9];
    {
        const float* ra = p.mag + ((long long)b * T + n) * (kN1 / 2 + 1);
        const float* rb = ra + (two ? (kN1 / 2 + 1) : 0);              // a lone last frame: b re-reads a (discarded)
        const float sb = two ? (1.0f / kM) : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = l ? (l + 128 * u) : (u < 4 ? 128 * u : 64 + 128 * (u - 4));
            const int kk = k <= 512 ? k : 1024 - k;
            ma[u] = ra[kk] * (1.0f / kM);                             // 1/M of the inverse transform (exact)
            mb[u] = rb[kk] * sb;
        }
        ma[8] = ra[512] * (1.0f / kM);
        mb[8] = rb[512] * sb;
    }

    v2f z[16];
    if (p.x_in != nullptr) {
        const float* xr = p.x_in + (long long)b * p.L + (long long)n * H;
        const int hb = two ? H : 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            z[j].x = xr[l + 64 * j];
            z[j].y = xr[hb + l + 64 * j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) z[j].y = two ? z[j].y : 0.0f;
    } else {
        const float* yb = p.y_in + (long long)b * T * kN1;
        // Sample i of frame a sits at s = nH + i of the signal; the frames covering s are m = m_hi - reach + c,
        // c = 0..reach (ascending), m_hi = floor(s / H) = n + floor(i / H), at position s - mH, which is inside
        // the frame when it is < N.  Frame b's sample i sits H further: same positions, frames one later.
        // Every load is issued (clamped address) and masked afterwards, so the 32 of one c are in flight at once.
        const float rh = 1.0f / (float)H;
        const int last = T - 1;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) z[jj] = (v2f){0.0f, 0.0f};
        // (Measured alternative: all loads of four contributions first, then the sums, with the register budget
        // of one wave per SIMD -- 12.5 instead of 13.5 us at B=1, but 24 instead of 19 us at B=4: the kernel then
        // needs a second round of workgroups from 2 048 frames on.)
#pragma unroll
        for (int c = 0; c <= REACH; ++c) {
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int i = l + 64 * jj;
                const int q = (int)(((float)i + 0.5f) * rh);          // floor(i / H), exact for i < 1024, H >= 114
                const int dm = q - REACH + c;                         // m - n for frame a
                const int pos = i - dm * H;                           // >= 0 by construction
                const int m_a = n + dm, m_b = m_a + 1;
                const bool in = pos < kN1;
                const float va = yb[(long long)min(max(m_a, 0), last) * kN1 + min(pos, kN1 - 1)];
                const float vb = yb[(long long)min(max(m_b, 0), last) * kN1 + min(pos, kN1 - 1)];
                z[jj].x += (in && m_a >= 0 && m_a <= last) ? va : 0.0f;
                z[jj].y += (two && in && m_b >= 0 && m_b <= last) ? vb : 0.0f;
            }
        }
    }
    bool a_nz = false, b_nz = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        a_nz |= (z[j].x != 0.0f);
        b_nz |= (z[j].y != 0.0f);
        z[j] = z[j] * tb.get(kT2Win + j);
    }
    // an all-zero input frame must give an exactly zero spectrum (see k_gl1024_pair)
    const float ka = __any(a_nz) ? 1.0f : 0.0f;
    const float kb = __any(b_nz) ? 1.0f : 0.0f;

    Stamps st;
    fft_forward_1024(z, ex, l, a3, g3, kloB, tb, st);
    middle_two_frames(z, l0, ma, mb, ka, kb);
    fft_inverse_1024(z, ex, l, a3, g3, kloB, tb, st);

    float* ya = p.y_out + ((long long)b * T + n) * kN1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const v2f w = tb.get(kT2Win + j);
        ya[l + 64 * j] = z[j].x * w.x;
        if (two) ya[kN1 + l + 64 * j] = z[j].y * w.y;
    }
}

// x[b][run*G*H + i] += tail[b][run-1][i], i < N-H, run = 1..nruns-1, in place: completes the signal between two
// iterations of the n_fft-1024 run kernels (one add per sample, the same "main + tail" the consumer used to do).
__global__ void __launch_bounds__(256) k_fold_seams(float* __restrict__ x, const float* __restrict__ tail, long long L,
                                                    int g_h, int nruns, int ovl, int blocks_per_seam) {
    const long long seam = blockIdx.x / blocks_per_seam;             // (b, run - 1), flattened
    const int i = (int)(blockIdx.x - seam * blocks_per_seam) * 256 + threadIdx.x;
    const int b = (int)(seam / (nruns - 1));
    const int run = (int)(seam - (long long)b * (nruns - 1)) + 1;
    if (i >= ovl) return;
    const long long s = (long long)run * g_h + i;
    if (s >= L) return;
    x[(long long)b * L + s] += tail[((long long)b * nruns + run - 1) * ovl + i];
}

// x_out[s] = x_main[s] + tail of the previous run (final fold after the last iteration)
__global__ void __launch_bounds__(256) k_fold_tails(const float* __restrict__ x_main, const float* __restrict__ tail,
                                                    float* __restrict__ x_out, long long L, int g_h, int nruns,
                                                    int ovl) {
    const int b = blockIdx.y;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    x_out[(long long)b * L + s] =
        fetch_x(x_main + (long long)b * L, tail + (long long)b * nruns * ovl, s, g_h, nruns, ovl);
}

// Post pass of the optional (non-reference) variants, one launch per iteration when enabled:
// v = x_main[s] + tail; window-sum normalisation v *= inv_wsum[s]; fast Griffin-Lim momentum
// t = v + alpha (v - prev[s]), prev[s] = v (Perraudin et al. 2013, in the signal domain: the STFT
// is linear, so extrapolating the signals extrapolates the consistent spectrograms).  The first
// iteration has no predecessor and the last one returns v itself.
__global__ void __launch_bounds__(256) k_post_iter(const float* __restrict__ x_main, const float* __restrict__ tail,
                                                   const float* __restrict__ inv_wsum, float* __restrict__ prev,
                                                   float* __restrict__ out, long long L, int g_h, int nruns, int ovl,
                                                   float alpha, int first, int last) {
    const int b = blockIdx.y;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    float v = fetch_x(x_main + (long long)b * L, tail ? tail + (long long)b * nruns * ovl : nullptr, s, g_h, nruns, ovl);
    if (inv_wsum) v *= inv_wsum[s];
    float t = v;
    if (prev) {
        if (!first && !last) t = v + alpha * (v - prev[(long long)b * L + s]);
        if (!last) prev[(long long)b * L + s] = v;
    }
    out[(long long)b * L + s] = t;
}

// Frames per wave-run.  A run costs its frames plus a fixed prologue/epilogue (ring fill,
// tail write, ~1.5 frame-times); the launch finishes after ceil(runs / resident waves)
// rounds, so choose the G that minimises rounds * (G + overhead).  G*H >= N-H keeps the
// number of runs that touch one sample at two.
int pick_run_length(const gl_plan* p, int B, int T, int waves_per_cu) {
    const int N = p->n_fft, H = p->hop;
    int g_min = (N - H + H - 1) / H;
    if (g_min < 1) g_min = 1;
    int pinned = p->run_frames;
    if (const char* e = getenv("GL_RUN_FRAMES")) pinned = atoi(e);   // experiments
    if (pinned > 0) {
        int G = pinned < g_min ? g_min : pinned;
        return G > T ? T : G;
    }
    if (T <= g_min) return T;
    const long long slots = (long long)p->n_cu * waves_per_cu;
    int best = g_min;
    double best_cost = 1e300;
    const int g_max = T < 96 ? T : 96;
    for (int G = g_min; G <= g_max; ++G) {
        const long long runs = (long long)B * ((T + G - 1) / G);
        const long long rounds = (runs + slots - 1) / slots;
        const double cost = (double)rounds * (G + 1.5);
        if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && G > best)) {
            best_cost = cost;
            best = G;
        }
    }
    return best;
}

int ensure_table(gl_plan* p) {
    if (p->d_fast_tab) return 0;
    const int N = p->n_fft;
    std::vector<float> tab((size_t)kTabEntries * 64, 0.0f);
    const double pi = 3.14159265358979323846;
    auto W = [&](int n, long long k, float* re, float* im) {
        double a = -2.0 * pi * (double)(k % n) / (double)n;
        *re = (float)cos(a);
        *im = (float)sin(a);
    };
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 16; ++j)
            for (int e = 0; e < 2; ++e) {
                int n = 2 * (l + 64 * j) + e;
                tab[(size_t)(kTabWin + 2 * j + e) * 64 + l] = (float)(0.5 - 0.5 * cos(2.0 * pi * n / (double)(N - 1)));
            }
        for (int k2 = 1; k2 < 16; ++k2)
            W(1024, (long long)l * k2, &tab[(size_t)(kTabTw1 + 2 * (k2 - 1)) * 64 + l],
              &tab[(size_t)(kTabTw1 + 2 * (k2 - 1) + 1) * 64 + l]);
        for (int d = 1; d < 8; ++d)
            W(64, (long long)(l & 7) * d, &tab[(size_t)(kTabTw2 + 2 * (d - 1)) * 64 + l],
              &tab[(size_t)(kTabTw2 + 2 * (d - 1) + 1) * 64 + l]);
        for (int u = 0; u < 9; ++u) {
            int k = (u == 8) ? 512 : (l ? l + 128 * u : (u < 4 ? 128 * u : 64 + 128 * (u - 4)));
            W(2048, k, &tab[(size_t)(kTabTwp + 2 * u) * 64 + l], &tab[(size_t)(kTabTwp + 2 * u + 1) * 64 + l]);
        }
    }
    int rc = gl_plan_alloc(p, (void**)&p->d_fast_tab, tab.size() * sizeof(float));
    if (rc) return rc;
    GL_HIP_CHECK(hipMemcpy(p->d_fast_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int ensure_buffers(gl_plan* p) {
    const int N = p->n_fft, H = p->hop;
    size_t xbytes = (size_t)p->max_B * gl_len(N, H, p->max_T) * sizeof(float);
    int g_min = (N - H + H - 1) / H;
    if (g_min < 1) g_min = 1;
    size_t max_runs = (size_t)(p->max_T + g_min - 1) / g_min;
    size_t tbytes = (size_t)p->max_B * max_runs * (size_t)(N - H) * sizeof(float);
    int rc = 0;
    if (!p->d_xa) rc = gl_plan_alloc(p, (void**)&p->d_xa, xbytes);
    if (!rc && !p->d_xb) rc = gl_plan_alloc(p, (void**)&p->d_xb, xbytes);
    if (!rc && !p->d_tail_a) rc = gl_plan_alloc(p, (void**)&p->d_tail_a, tbytes);
    if (!rc && !p->d_tail_b) rc = gl_plan_alloc(p, (void**)&p->d_tail_b, tbytes);
    return rc;
}

// =====================================================================================
// N3 forward analysis, n_fft 2048 (data.py:110-111: np.abs(librosa.stft(y, n_fft))): centred frames at any
// hop, periodic Hann, |rfft|.  One wave per frame on the register FFT core (same passes and mirror-pair
// split as the Griffin-Lim kernels), frames dealt to the waves of one resident round; the centre padding
// (reflect, or zeros) is folded into the frame gather of the few edge frames.
struct AnalysisArgs {
    const float* y;          // [B][y_stride], len valid samples each
    long long y_stride, len;
    float* mag;              // [B][T][1025]
    const v2f* tab;          // [kT2Entries][64]
    const float* win_p;      // periodic Hann, 2048 floats
    int H, T, reflect;
    long long n_frames;      // B * T
};
typedef float v2f_u __attribute__((ext_vector_type(2), aligned(4)));   // frame starts are only 4-byte aligned

__global__ void __launch_bounds__(64 * kRRWaves) k_analysis2048(AnalysisArgs p) {
    __shared__ __attribute__((aligned(16))) v2f smem[kT2Entries * 64 + kRRWaves * kRScratch];
    const int l = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < kT2Entries * 64; i += 64 * kRRWaves) smem[i] = p.tab[i];
    __syncthreads();                             // no barrier after this point
    v2f* ex = smem + kT2Entries * 64 + wave * kRScratch;
    const TabLds tb{smem + 2 * l};
    const int a3 = l & 7, g3 = l >> 3;
    const int kloB = l ? 128 - l : 64;
    const bool l0 = (l == 0);
    v2f win[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) win[j] = *reinterpret_cast<const v2f*>(p.win_p + 2 * (l + 64 * j));
    Stamps st;
    st.start();
    const long long step = (long long)gridDim.x * kRRWaves;
    for (long long fr = (long long)blockIdx.x * kRRWaves + wave; fr < p.n_frames; fr += step) {
        const int b = (int)(fr / p.T);
        const int t = (int)(fr - (long long)b * p.T);
        const float* yr = p.y + (long long)b * p.y_stride;
        const long long s0 = (long long)t * p.H - kM;
        v2f z[16];
        if (s0 >= 0 && s0 + kN <= p.len) {       // wave-uniform: the frame lies inside the signal
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2f_u v = *reinterpret_cast<const v2f_u*>(yr + s0 + 2 * (l + 64 * j));
                z[j] = (v2f){v.x, v.y} * win[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    long long sidx = s0 + 2 * (l + 64 * j) + e;
                    bool inside = (sidx >= 0 && sidx < p.len);
                    if (!inside && p.reflect) {
                        sidx = sidx < 0 ? -sidx : 2 * (p.len - 1) - sidx;
                        inside = (sidx >= 0 && sidx < p.len);
                    }
                    v[e] = inside ? yr[sidx] : 0.0f;
                }
                z[j] = (v2f){v[0], v[1]} * win[j];
            }
        }
        fft_forward_1024(z, ex, l, a3, g3, kloB, tb, st);
        // mirror-pair split (see middle_project): 2 X[k] = (P + conj Q) + w (P - conj Q), 2 conj X[M-k] = ... - ...
        float* orow = p.mag + fr * (kM + 1);
        const v2f z4 = z[4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v2f P, Q;
            if (u == 0) { P = z[0]; Q = sel(l0, z[0], z[15]); }
            else if (u < 4) { P = z[u]; Q = sel(l0, z[8 - u], z[15 - u]); }
            else { P = sel(l0, z[8 + u - 4], z[u]); Q = sel(l0, z[8 + 11 - u], z[15 - u]); }
            const v2f w = tb.get(kT2Twp + u);
            const v2f uu = pk_addc(P, Q), vv = pk_subc(P, Q);
            const v2f X = cmul_add(vv, w, uu), Y = cmul_sub(vv, w, uu);
            const int k = l ? l + 128 * u : (u < 4 ? 128 * u : 64 + 128 * (u - 4));
            orow[k] = 0.5f * sqrtf(fmaf(X.x, X.x, X.y * X.y));
            orow[kM - k] = 0.5f * sqrtf(fmaf(Y.x, Y.x, Y.y * Y.y));
        }
        if (l0) orow[512] = sqrtf(fmaf(z4.x, z4.x, z4.y * z4.y));   // bin 512 is its own mirror
    }
    st.flush(l);
}

template <int MODE>
int launch(const FastArgs& a, int B, hipStream_t st) {
    dim3 grid(a.nruns, B);
    hipLaunchKernelGGL(k_fast2048<MODE>, grid, dim3(64), 0, st, a);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int fold(gl_plan* p, const float* x_main, const float* tail, float* x_out, int B, int T, int G, int nruns,
         hipStream_t st) {
    const long long L = gl_len(p->n_fft, p->hop, T);
    dim3 grid((unsigned)((L + 255) / 256), B);
    hipLaunchKernelGGL(k_fold_tails, grid, dim3(256), 0, st, x_main, tail, x_out, L, G * p->hop, nruns,
                       p->n_fft - p->hop);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}


int ensure_table2(gl_plan* p) {
    if (p->d_fast_tab2) return 0;
    const int N = p->n_fft;
    std::vector<float> tab((size_t)(kT2Zero + 1) * 64 * 2, 0.0f);
    const double pi = 3.14159265358979323846;
    auto put = [&](int e, int l, double re, double im) {      // pair layout, see TE()
        const size_t at = ((size_t)(e >> 1) * 128 + 2 * (size_t)l + (e & 1)) * 2;
        tab[at] = (float)re;
        tab[at + 1] = (float)im;
    };
    auto Wre = [&](int n, long long k) { return cos(-2.0 * pi * (double)(k % n) / (double)n); };
    auto Wim = [&](int n, long long k) { return sin(-2.0 * pi * (double)(k % n) / (double)n); };
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 16; ++j) {
            // n_fft 2048: lane holds the even/odd sample pair 2(l+64j), +1 of ONE frame;
            // n_fft 1024: lane holds sample l+64j of TWO frames (same window value twice)
            int n0 = (N == kN) ? 2 * (l + 64 * j) : (l + 64 * j);
            int n1 = (N == kN) ? n0 + 1 : n0;
            float w0 = (float)(0.5 - 0.5 * cos(2.0 * pi * n0 / (double)(N - 1)));
            float w1 = (float)(0.5 - 0.5 * cos(2.0 * pi * n1 / (double)(N - 1)));
            put(kT2Win + j, l, w0, w1);
            put(kT2WinSc + j, l, (double)w0 / kM, (double)w1 / kM);   // exact: power-of-two scale (n_fft 1024)
        }
        for (int k2 = 1; k2 < 16; ++k2) put(kT2Tw1 + k2 - 1, l, Wre(1024, (long long)l * k2), Wim(1024, (long long)l * k2));
        for (int d = 1; d < 8; ++d) put(kT2Tw2 + d - 1, l, Wre(64, (long long)(l & 7) * d), Wim(64, (long long)(l & 7) * d));
        for (int u = 0; u < 9; ++u) {
            int k = (u == 8) ? 512 : (l ? l + 128 * u : (u < 4 ? 128 * u : 64 + 128 * (u - 4)));
            put(kT2Twp + u, l, Wim(2048, k), -Wre(2048, k));   // -i * W2048^k (middle_project)
        }
        // DFT-16 as a 32x32 real matrix (dft16_mfma): lane (i = l&31, kk = l>>5) supplies A[i][2s+kk] for the
        // K-step s = input index; output row i is component i&1 (re, im) of output k2 = i>>1
        {
            const int i = l & 31, kk = l >> 5, k2 = i >> 1, comp = i & 1;
            for (int s2 = 0; s2 < 16; ++s2) {
                const double th = 2.0 * pi * (double)((s2 * k2) % 16) / 16.0;
                const double c = cos(th), sn = sin(th);
                // forward e^{-i th}: re' = c re + sn im, im' = -sn re + c im; inverse e^{+i th}: signs of sn flipped
                const double af = comp == 0 ? (kk == 0 ? c : sn) : (kk == 0 ? -sn : c);
                const double ai = comp == 0 ? (kk == 0 ? c : -sn) : (kk == 0 ? sn : c);
                const size_t ef = kT2A16F + s2 / 2, ei = kT2A16I + s2 / 2;
                tab[((size_t)(ef >> 1) * 128 + 2 * (size_t)l + (ef & 1)) * 2 + (s2 & 1)] = (float)af;
                tab[((size_t)(ei >> 1) * 128 + 2 * (size_t)l + (ei & 1)) * 2 + (s2 & 1)] = (float)ai;
            }
        }
    }
    int rc = gl_plan_alloc(p, (void**)&p->d_fast_tab2, tab.size() * sizeof(float));
    if (rc) return rc;
    GL_HIP_CHECK(hipMemcpy(p->d_fast_tab2, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

bool regring_ok(const gl_plan* p, const void* a, const void* b) {
    const int H = p->hop;
    if (H % 128 != 0) return false;
    const int hs = H / 128;
    if (hs != 1 && hs != 2 && hs != 4 && hs != 8) return false;
    return (((uintptr_t)a | (uintptr_t)b) & 7) == 0;
}

int launch_regring(const gl_plan* p, const AlignedArgs& a, int B, hipStream_t st) {
    dim3 grid((unsigned)(((long long)a.nruns * B + kRRWaves - 1) / kRRWaves));
    dim3 block(64 * kRRWaves);
    static const size_t pad = getenv("GL_RR_PADLDS") ? (size_t)atoi(getenv("GL_RR_PADLDS")) : 0;  // experiments
    switch (p->hop / 128) {
        case 1: hipLaunchKernelGGL(k_gl2048_regring<1>, grid, block, pad, st, a); break;
        case 2: hipLaunchKernelGGL(k_gl2048_regring<2>, grid, block, pad, st, a); break;
        case 4: hipLaunchKernelGGL(k_gl2048_regring<4>, grid, block, pad, st, a); break;
        default: hipLaunchKernelGGL(k_gl2048_regring<8>, grid, block, pad, st, a); break;
    }
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

// Frames (B*T) up to which the per-frame form of n_fft 2048 is used (experiments: GL_F2_MAX_FRAMES).  Measured at
// hop 256, T = 861 (us per iteration, per-frame form vs register-ring runs): B=1 22.0 / 34.6, B=2 23.4 / 35.0,
// B=4 45.5 / 35.9, one 6 600-frame chunk 89.7 / 37.1: the run kernel scales far better (no 8-fold gather), so
// the per-frame form is for one or two utterances only.
static long long kF2MaxFrames(const gl_plan* p) {
    static const char* e = getenv("GL_F2_MAX_FRAMES");
    return e ? atoll(e) : 8LL * p->n_cu;
}

int run_regring(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                hipStream_t st) {
    int rc = ensure_table2(p);
    if (!rc) rc = ensure_buffers(p);
    if (rc) return rc;
    const int G = pick_run_length(p, B, T, GL_RR_OCC * kRRWaves);
    const int nruns = (T + G - 1) / G;
    if (!p->d_magp) {
        rc = gl_plan_alloc(p, (void**)&p->d_magp, (size_t)p->max_B * p->max_T * kMagPack * sizeof(float));
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_mag_repack, dim3(T, B), dim3(64), 0, st, mag, p->d_magp, T);
    GL_HIP_CHECK(hipGetLastError());
    // Too few frames to fill the GPU with runs: one wave per frame, state kept as frames (k_gl2048_frames).  Not
    // with a pinned run length (callers who pin it compare bits with the run kernel) nor with the optional
    // per-iteration signal transforms.
    if (!gl_post_needed(p) && p->run_frames == 0 && !getenv("GL_RUN_FRAMES") && iters >= 1 && p->hop >= 256 && p->hop < kN &&
        (long long)B * T <= kF2MaxFrames(p)) {
        const size_t need = (size_t)2 * B * T * kN * sizeof(float);
        if (p->yframes_bytes < need) {
            // grown on demand, never under stream capture: gl_fast_prepare sizes it for the plan's maximum
            if (p->d_yframes) { GL_HIP_CHECK(hipStreamSynchronize(st)); GL_HIP_CHECK(hipFree(p->d_yframes)); p->bytes -= (long long)p->yframes_bytes; p->d_yframes = nullptr; p->yframes_bytes = 0; }
            rc = gl_plan_alloc(p, (void**)&p->d_yframes, need);
            if (rc) return rc;
            p->yframes_bytes = need;
        }
        Frames2048Args f = {};
        f.magp = p->d_magp;
        f.tab = reinterpret_cast<const v2f*>(p->d_fast_tab2);
        f.T = T;
        f.B = B;
        f.L = gl_len(p->n_fft, p->hop, T);
        float* ybuf[2] = {p->d_yframes, p->d_yframes + (size_t)B * T * kN};
        const dim3 grid((unsigned)(((long long)B * T + kRRWaves - 1) / kRRWaves)), block(64 * kRRWaves);
        for (int it = 0; it < iters; ++it) {
            f.x_in = it == 0 ? x0 : nullptr;
            f.y_in = it == 0 ? nullptr : ybuf[(it - 1) & 1];
            f.y_out = ybuf[it & 1];
            switch (p->hop / 128) {
                case 2: hipLaunchKernelGGL(k_gl2048_frames<2>, grid, block, 0, st, f); break;
                case 4: hipLaunchKernelGGL(k_gl2048_frames<4>, grid, block, 0, st, f); break;
                default: hipLaunchKernelGGL(k_gl2048_frames<8>, grid, block, 0, st, f); break;
            }
            GL_HIP_CHECK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_ola_frames, dim3((unsigned)((f.L + 255) / 256), B), dim3(256), 0, st, ybuf[(iters - 1) & 1],
                           x_out, f.L, kN, p->hop, T);
        GL_HIP_CHECK(hipGetLastError());
        return 0;
    }
    AlignedArgs a = {};
    a.magp = p->d_magp;
    a.tab = reinterpret_cast<const v2f*>(p->d_fast_tab2);
    a.T = T;
    a.G = G;
    a.nruns = nruns;
    a.B = B;
    a.L = gl_len(p->n_fft, p->hop, T);
    const float* src = x0;
    const float* src_tail = nullptr;
    const bool post = gl_post_needed(p);
    for (int it = 0; it < iters; ++it) {
        const bool last = (it == iters - 1);
        float* dst = (it & 1) ? p->d_xb : p->d_xa;
        float* dst_tail = (it & 1) ? p->d_tail_b : p->d_tail_a;
        if (last && nruns == 1) dst = x_out;
        a.x_in = src;
        a.tail_in = src_tail;
        a.x_out = dst;
        a.tail_out = dst_tail;
        rc = launch_regring(p, a, B, st);
        if (rc) return rc;
        src = dst;
        src_tail = (nruns > 1) ? dst_tail : nullptr;
        if (post) {      // optional variants: complete the signal now and transform it
            float* o = last ? x_out : dst;
            rc = gl_post_iter(p, dst, src_tail, o, B, T, G, nruns, it == 0, last, st);
            if (rc) return rc;
            src = o;
            src_tail = nullptr;
        }
    }
#ifdef GL_STAMPS
    {
        unsigned long long h[24];
        GL_HIP_CHECK(hipStreamSynchronize(st));
        GL_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h)));
        const double frames = (double)B * T * iters;
        double tot = 0;
        for (int i = 0; i < 16; ++i) tot += (double)h[i];
        fprintf(stderr, "[GL_STAMPS] shader-clock ticks per frame per wave (G=%d, waves %llu): total %.0f\n", G, h[20], tot / frames);
        for (int i = 0; i < 16; ++i) fprintf(stderr, "  phase %2d: %8.1f\n", i, (double)h[i] / frames);
        memset(h, 0, sizeof(h));
        GL_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), h, sizeof(h)));
    }
#endif
    if (nruns > 1 && !post) return fold(p, src, src_tail, x_out, B, T, G, nruns, st);
    return 0;
}


// Frames (B*T) up to which the per-pair form is used.  Measured at T = 641, hop 275 (us per iteration, per-pair
// form vs the run kernel k_gl1024_pair2): B=1 13.5 / 20.0, B=4 19.3 / 23.8, B=6 20.6 / 25.2, B=8 24.9 / 25.4,
// B=12 33.4 / 27.8, B=16 42.7 / 33.6: level at about 5 000 frames.
static long long kF1MaxFrames(const gl_plan* p) {
    static const char* e = getenv("GL_F1_MAX_FRAMES");   // experiments
    return e ? atoll(e) : 20LL * p->n_cu;
}

int run_pair1024(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                 hipStream_t st) {
    int rc = ensure_table2(p);
    if (!rc) rc = ensure_buffers(p);
    if (rc) return rc;
    // Three forms: the per-pair frames form for small problems (below), k_gl1024_pair2 for everything else, and the
    // first run kernel k_gl1024_pair (4 waves per CU, x ring in LDS), which nothing selects any more since the run
    // tails are folded between iterations (B=1, T=641: 24.9 us per iteration vs 20.0 for pair2 and 13.5 per-pair);
    // it stays as an independent implementation the tests force with GL_P1_VERSION=1.
    static const char* force = getenv("GL_P1_VERSION");        // experiments: "1", "2" or "f" (per-pair frames form)
    const bool post = gl_post_needed(p);
    // Too few frames to fill the GPU with runs: one wave per frame pair, state kept as frames (k_gl1024_frames).
    // Not with a pinned run length (callers who pin it compare bits with the run kernels) nor with the optional
    // per-iteration signal transforms, and only while a sample's reach stays small.
    const int reach = (kN1 - 1) / p->hop;
    const bool frames_ok = !post && p->run_frames == 0 && !getenv("GL_RUN_FRAMES") && reach >= 1 && reach <= kF1MaxReach && iters >= 1;
    const bool frames_form = force ? (force[0] == 'f' && frames_ok) : (frames_ok && (long long)B * T <= kF1MaxFrames(p));
    if (frames_form) {
        const size_t need = (size_t)2 * B * T * kN1 * sizeof(float);
        if (p->yframes_bytes < need) {
            // grown on demand, never under stream capture: gl_fast_prepare sizes it for the plan's maximum
            if (p->d_yframes) { GL_HIP_CHECK(hipStreamSynchronize(st)); GL_HIP_CHECK(hipFree(p->d_yframes)); p->bytes -= (long long)p->yframes_bytes; p->d_yframes = nullptr; p->yframes_bytes = 0; }
            rc = gl_plan_alloc(p, (void**)&p->d_yframes, need);
            if (rc) return rc;
            p->yframes_bytes = need;
        }
        Frames1024Args f = {};
        f.mag = mag;
        f.tab = reinterpret_cast<const v2f*>(p->d_fast_tab2);
        f.H = p->hop;
        f.T = T;
        f.B = B;
        f.reach = reach;
        f.L = gl_len(p->n_fft, p->hop, T);
        float* ybuf[2] = {p->d_yframes, p->d_yframes + (size_t)B * T * kN1};
        const long long pairs = (long long)B * ((T + 1) / 2);
        for (int it = 0; it < iters; ++it) {
            f.x_in = it == 0 ? x0 : nullptr;
            f.y_in = it == 0 ? nullptr : ybuf[(it - 1) & 1];
            f.y_out = ybuf[it & 1];
            const dim3 grid((unsigned)((pairs + kF1Waves - 1) / kF1Waves)), block(64 * kF1Waves);
            switch (reach) {
                case 1: hipLaunchKernelGGL(k_gl1024_frames<1>, grid, block, 0, st, f); break;
                case 2: hipLaunchKernelGGL(k_gl1024_frames<2>, grid, block, 0, st, f); break;
                case 3: hipLaunchKernelGGL(k_gl1024_frames<3>, grid, block, 0, st, f); break;
                case 4: hipLaunchKernelGGL(k_gl1024_frames<4>, grid, block, 0, st, f); break;
                case 5: hipLaunchKernelGGL(k_gl1024_frames<5>, grid, block, 0, st, f); break;
                case 6: hipLaunchKernelGGL(k_gl1024_frames<6>, grid, block, 0, st, f); break;
                case 7: hipLaunchKernelGGL(k_gl1024_frames<7>, grid, block, 0, st, f); break;
                default: hipLaunchKernelGGL(k_gl1024_frames<8>, grid, block, 0, st, f); break;
            }
            GL_HIP_CHECK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_ola_frames, dim3((unsigned)((f.L + 255) / 256), B), dim3(256), 0, st, ybuf[(iters - 1) & 1],
                           x_out, f.L, kN1, p->hop, T);
        GL_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const bool v1 = force && force[0] == '1';
    const int waves = v1 ? kP1Waves : kP2Waves;
    const int G = pick_run_length(p, B, T, waves);
    const int nruns = (T + G - 1) / G;
    Pair2Args aa = {};
    Pair1024Args& a = aa.a;
    aa.B = B;
    a.mag = mag;
    a.tab = reinterpret_cast<const v2f*>(p->d_fast_tab2);
    a.H = p->hop;
    a.T = T;
    a.G = G;
    a.nruns = nruns;
    a.L = gl_len(p->n_fft, p->hop, T);
    if (!v1) {
        // k_gl1024_pair2: seams by atomics onto cleared regions; three signal buffers rotate (read, write, clear)
        if (!p->d_xc) {
            rc = gl_plan_alloc(p, (void**)&p->d_xc, (size_t)p->max_B * gl_len(p->n_fft, p->hop, p->max_T) * sizeof(float));
            if (rc) return rc;
        }
        float* buf[3] = {p->d_xa, p->d_xb, p->d_xc};
        const size_t bytes = (size_t)B * a.L * sizeof(float);
        if (nruns > 1) GL_HIP_CHECK(hipMemsetAsync(buf[0], 0, bytes, st));     // seams of the first target
        const float* src = x0;
        for (int it = 0; it < iters; ++it) {
            const bool last = (it == iters - 1);
            float* dst = buf[it % 3];
            a.x_in = src;
            a.x_out = dst;
            aa.x_zero = (nruns > 1 && !last) ? buf[(it + 1) % 3] : nullptr;
            hipLaunchKernelGGL(k_gl1024_pair2, dim3((unsigned)(((long long)nruns * B + kP2Waves - 1) / kP2Waves)),
                               dim3(64 * kP2Waves), 0, st, aa);
            GL_HIP_CHECK(hipGetLastError());
            src = dst;
            if (post) {      // optional variants: the signal is complete; transform it (in place, or into x_out)
                float* o = last ? x_out : dst;
                rc = gl_post_iter(p, dst, nullptr, o, B, T, G, nruns, it == 0, last, st);
                if (rc) return rc;
                src = o;
            }
        }
        if (!post) GL_HIP_CHECK(hipMemcpyAsync(x_out, src, bytes, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    // first version (forced by the tests): tails in a side buffer, folded in place between iterations
    const float* src = x0;
    const float* src_tail = nullptr;
    for (int it = 0; it < iters; ++it) {
        const bool last = (it == iters - 1);
        float* dst = (it & 1) ? p->d_xb : p->d_xa;
        float* dst_tail = (it & 1) ? p->d_tail_b : p->d_tail_a;
        if (last && nruns == 1) dst = x_out;
        a.x_in = src;
        a.tail_in = src_tail;
        a.x_out = dst;
        a.tail_out = dst_tail;
        hipLaunchKernelGGL(k_gl1024_pair, dim3((nruns + kP1Waves - 1) / kP1Waves, B), dim3(64 * kP1Waves), 0, st, a);
        GL_HIP_CHECK(hipGetLastError());
        src = dst;
        src_tail = (nruns > 1) ? dst_tail : nullptr;
        if (post) {      // optional variants: complete the signal now and transform it
            float* o = last ? x_out : dst;
            rc = gl_post_iter(p, dst, src_tail, o, B, T, G, nruns, it == 0, last, st);
            if (rc) return rc;
            src = o;
            src_tail = nullptr;
        } else if (!last && nruns > 1) {
            // complete the signal in place: the kernels read x_in without tails
            const int ovl = p->n_fft - p->hop;
            const int bps = (ovl + 255) / 256;
            if (ovl > 0) {                                            // hop == n_fft: frames do not overlap, no tails
                hipLaunchKernelGGL(k_fold_seams, dim3((unsigned)((long long)bps * (nruns - 1) * B)), dim3(256), 0, st,
                                   dst, dst_tail, a.L, G * p->hop, nruns, ovl, bps);
                GL_HIP_CHECK(hipGetLastError());
            }
            src_tail = nullptr;
        }
    }
    if (nruns > 1 && !post) return fold(p, src, src_tail, x_out, B, T, G, nruns, st);
    return 0;
}

}  // namespace

int gl_post_iter(gl_plan* p, const float* x_main, const float* tail, float* out, int B, int T, int G, int nruns,
                 bool first, bool last, hipStream_t st) {
    const long long L = gl_len(p->n_fft, p->hop, T);
    dim3 grid((unsigned)((L + 255) / 256), B);
    hipLaunchKernelGGL(k_post_iter, grid, dim3(256), 0, st, x_main, tail,
                       (p->opt_flags & GL_OPT_WSUM_NORMALISE) ? p->d_inv_wsum : nullptr,
                       p->momentum != 0.0f ? p->d_prev : nullptr, out, L, G * p->hop, nruns, p->n_fft - p->hop,
                       p->momentum, first ? 1 : 0, last ? 1 : 0);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

bool gl_fast_supported(const gl_plan* p, int op) {
    if (p->hop < 1) return false;
    if (p->n_fft == kN && p->hop <= kN) return true;                          // all three entry points
    if (p->n_fft == kN1 && p->hop <= kN1) return op == GL_FAST_OP_RUN;         // Griffin-Lim loop only
    return false;
}

// every lazy allocation / table upload gl_fast_run could need, so that a later call only enqueues
int gl_fast_prepare(gl_plan* p) {
    int rc = ensure_buffers(p);
    if (rc) return rc;
    if (p->n_fft == kN1) {
        rc = ensure_table2(p);
        if (!rc && !p->d_xc)
            rc = gl_plan_alloc(p, (void**)&p->d_xc, (size_t)p->max_B * gl_len(p->n_fft, p->hop, p->max_T) * sizeof(float));
        long long fr = (long long)p->max_B * p->max_T;
        if (fr > kF1MaxFrames(p)) fr = kF1MaxFrames(p);
        const size_t need = (size_t)2 * fr * kN1 * sizeof(float);          // per-pair form of small problems
        if (!rc && p->yframes_bytes < need) {
            if (p->d_yframes) { GL_HIP_CHECK(hipFree(p->d_yframes)); p->bytes -= (long long)p->yframes_bytes; p->d_yframes = nullptr; p->yframes_bytes = 0; }
            rc = gl_plan_alloc(p, (void**)&p->d_yframes, need);
            if (!rc) p->yframes_bytes = need;
        }
        return rc;
    }
    rc = ensure_table(p);
    if (!rc && p->hop % 128 == 0) {
        rc = ensure_table2(p);
        if (!rc && !p->d_magp)
            rc = gl_plan_alloc(p, (void**)&p->d_magp, (size_t)p->max_B * p->max_T * kMagPack * sizeof(float));
        long long fr = (long long)p->max_B * p->max_T;
        if (fr > kF2MaxFrames(p)) fr = kF2MaxFrames(p);
        const size_t need = (size_t)2 * fr * kN * sizeof(float);              // per-frame form of small problems
        if (!rc && p->hop >= 256 && p->yframes_bytes < need) {
            if (p->d_yframes) { GL_HIP_CHECK(hipFree(p->d_yframes)); p->bytes -= (long long)p->yframes_bytes; p->d_yframes = nullptr; p->yframes_bytes = 0; }
            rc = gl_plan_alloc(p, (void**)&p->d_yframes, need);
            if (!rc) p->yframes_bytes = need;
        }
    }
    return rc;
}

int gl_fast_stft(gl_plan* p, const float* x, long long x_stride, float* S, int B, int T, hipStream_t st) {
    int rc = ensure_table(p);
    if (rc) return rc;
    FastArgs a = {};
    a.x_in = x;
    a.x_stride = x_stride;
    a.S_out = (float2*)S;
    a.tab = p->d_fast_tab;
    a.H = p->hop;
    a.T = T;
    a.G = pick_run_length(p, B, T, 4);
    a.nruns = (T + a.G - 1) / a.G;
    a.L = gl_len(p->n_fft, p->hop, T);
    return launch<MODE_STFT>(a, B, st);
}

int gl_fast_istft(gl_plan* p, const float* S, float* y, int B, int T, hipStream_t st) {
    int rc = ensure_table(p);
    if (!rc) rc = ensure_buffers(p);
    if (rc) return rc;
    FastArgs a = {};
    a.S_in = (const float2*)S;
    a.tab = p->d_fast_tab;
    a.H = p->hop;
    a.T = T;
    a.G = pick_run_length(p, B, T, 4);
    a.nruns = (T + a.G - 1) / a.G;
    a.L = gl_len(p->n_fft, p->hop, T);
    a.x_out = (a.nruns > 1) ? p->d_xa : y;
    a.tail_out = p->d_tail_a;
    rc = launch<MODE_ISTFT>(a, B, st);
    if (rc || a.nruns == 1) return rc;
    return fold(p, p->d_xa, p->d_tail_a, y, B, T, a.G, a.nruns, st);
}

// N3: |centred STFT| of B signals (n_fft 2048 only; gl_analysis_mag is the any-N path)
int gl_fast_analysis_mag(gl_plan* p, const float* y, long long y_stride, long long len, float* mag, int reflect,
                         int B, int T, const float* win_p, hipStream_t st) {
    int rc = ensure_table2(p);
    if (rc) return rc;
    AnalysisArgs a = {};
    a.y = y;
    a.y_stride = y_stride;
    a.len = len;
    a.mag = mag;
    a.tab = reinterpret_cast<const v2f*>(p->d_fast_tab2);
    a.win_p = win_p;
    a.H = p->hop;
    a.T = T;
    a.reflect = reflect;
    a.n_frames = (long long)B * T;
    long long wgs = (a.n_frames + kRRWaves - 1) / kRRWaves;
    if (wgs > p->n_cu) wgs = p->n_cu;            // one resident round; the waves walk the frames
    hipLaunchKernelGGL(k_analysis2048, dim3((unsigned)wgs), dim3(64 * kRRWaves), 0, st, a);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_fast_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                hipStream_t st) {
    if (p->n_fft == kN1) return run_pair1024(p, mag, x0, x_out, iters, B, T, st);
    if (p->path == 0 && regring_ok(p, x0, x_out)) return run_regring(p, mag, x0, x_out, iters, B, T, st);
    int rc = ensure_table(p);
    if (!rc) rc = ensure_buffers(p);
    if (rc) return rc;
    const int G = pick_run_length(p, B, T, 4);
    const int nruns = (T + G - 1) / G;
    const long long L = gl_len(p->n_fft, p->hop, T);
    FastArgs a = {};
    a.mag = mag;
    a.tab = p->d_fast_tab;
    a.H = p->hop;
    a.T = T;
    a.G = G;
    a.nruns = nruns;
    a.L = L;
    a.g_in = G;
    a.nruns_in = nruns;
    const float* src = x0;
    const float* src_tail = nullptr;
    const bool post = gl_post_needed(p);
    for (int it = 0; it < iters; ++it) {
        const bool last = (it == iters - 1);
        float* dst = (it & 1) ? p->d_xb : p->d_xa;
        float* dst_tail = (it & 1) ? p->d_tail_b : p->d_tail_a;
        if (last && nruns == 1) dst = x_out;     // single run: no tails, write the result directly
        a.x_in = src;
        a.x_stride = L;
        a.tail_in = src_tail;
        a.x_out = dst;
        a.tail_out = dst_tail;
        rc = launch<MODE_GL>(a, B, st);
        if (rc) return rc;
        src = dst;
        src_tail = (nruns > 1) ? dst_tail : nullptr;
        if (post) {      // optional variants: complete the signal now and transform it
            float* o = last ? x_out : dst;
            rc = gl_post_iter(p, dst, src_tail, o, B, T, G, nruns, it == 0, last, st);
            if (rc) return rc;
            src = o;
            src_tail = nullptr;
        }
    }
    if (nruns > 1 && !post) return fold(p, src, src_tail, x_out, B, T, G, nruns, st);
    return 0;
}
