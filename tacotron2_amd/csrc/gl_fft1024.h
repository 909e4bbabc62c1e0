// Shared device code of the fused wave-per-frame Griffin-Lim kernels (gl_fast2048.hip, gl_fast1024.hip):
// the 1024-point complex FFT a 64-lane wave keeps in registers (16 points per lane: radix-16, radix-8,
// radix-8 with two LDS transposes), the lane-local real-FFT split + magnitude projection, and the per-lane
// constant table.  gfx950 (MI355X) only.
//
// The index decomposition n = a + 8b + 64 n2, k = k2 + 16 d + 128 c and the lane assignment of the last
// pass (lane l owns the radix-8 butterflies k_lo = l and k_lo = 128 - l) put every mirror pair (k, 1024-k)
// of the real-FFT split in ONE lane, so the projection needs no cross-lane traffic.  The maps are modelled
// and checked on the CPU in oracle/sim_fast2048.py.
//
// Replaces np.fft.rfft / np.angle / np.exp / np.fft.irfft of audio_utilities.py:91-92, :181-183, :112.
#pragma once
#include "gl_internal.h"
#include "gl_experiment.h"

namespace gl_fft {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kN = 2048;           // n_fft
constexpr int kM = 1024;           // complex points

// ---- complex helpers on v2f --------------------------------------------------------------
#include "gl_pk.h"   // cmul, cmulc, cmul_add, ... and the one-instruction packed forms

__device__ __forceinline__ v2f lds_rd(const v2f* q) { return *q; }
typedef float v4f_ __attribute__((ext_vector_type(4)));
// two table entries of a pair (TE layout) with one 16-byte read
__device__ __forceinline__ void lds_rd_pair(const v2f* q, v2f& a, v2f& b) {
    const v4f_ t = *reinterpret_cast<const v4f_*>(q);
    a = t.xy;
    b = t.zw;
}

template <bool INV>
__device__ __forceinline__ v2f ctw(v2f a, v2f w) { return INV ? cmulc(a, w) : cmul(a, w); }
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
// Magnitude projection, audio_utilities.py:181-183: mag * exp(1j*angle(X)) as X * (mag * rsq(|X|^2)).
__device__ __forceinline__ v2f project(v2f X, float m) {
    // select-free form: nudging the real part by 2^-60 turns an exactly zero bin into (2^-60, 0), whose
    // projection is (mag, 0) = mag * exp(1j * angle(0)) (audio_utilities.py:181-183) without a compare;
    // a bin that large enough to matter (|X| > 2^-36 ~ 1.5e-11) does not see the nudge in fp32
    const v2f Xe = X + (v2f){0x1p-60f, 0.0f};
    const float r2 = fmaxf(fmaf(Xe.x, Xe.x, Xe.y * Xe.y), 0x1p-124f);
    const float sc = m * __builtin_amdgcn_rsqf(r2);
    return Xe * (v2f){sc, sc};
}

// Two bins at once, the form the n_fft-2048 kernels use: m2 = (magnitude of X, magnitude of Y).  |X|^2 + 2^-126
// keeps the reciprocal square root finite without a compare or a max (a bin that small is exactly zero in
// practice and comes out as 0 * finite = 0); an exactly zero FRAME never gets here as zeros: the kernels plant
// kZeroNudge in its sample 0, which the symmetric Hann window zeroes anyway (w[0] = 0), so its spectrum is the
// real constant kZeroNudge > 0 in every bin and projects to (mag, 0) = mag * exp(1j * np.angle(0)),
// audio_utilities.py:181-183.  The two scale factors are one packed multiply.
constexpr float kZeroNudge = 0x1p-52f;
__device__ __forceinline__ void project2(v2f& X, v2f& Y, v2f m2) {
    const float rx = fmaf(X.x, X.x, fmaf(X.y, X.y, 0x1p-126f));
    const float ry = fmaf(Y.x, Y.x, fmaf(Y.y, Y.y, 0x1p-126f));
    v2f rs;
    rs.x = __builtin_amdgcn_rsqf(rx);
    rs.y = __builtin_amdgcn_rsqf(ry);
#if defined(GL_EXPERIMENT) && defined(GL_PAR)
    // parity attribution builds (abl/build.sh par* -DGL_EXPERIMENT -DGL_PAR=n; tools_parity_distribution.py --lib):
    // 1 = one Newton step on v_rsq_f32 (1 ulp -> correctly rounded to within the step's own rounding)
    // 2 = IEEE square root and division instead of v_rsq_f32
    if (GL_PAR & 1) {
        rs.x = rs.x * fmaf(-0.5f * rx, rs.x * rs.x, 1.5f);
        rs.y = rs.y * fmaf(-0.5f * ry, rs.y * rs.y, 1.5f);
    }
    if (GL_PAR & 2) {
        rs.x = __fdiv_rn(1.0f, __fsqrt_rn(rx));
        rs.y = __fdiv_rn(1.0f, __fsqrt_rn(ry));
    }
#endif
    const v2f sc = m2 * rs;
    X = X * sc.xx;
    Y = Y * sc.yy;
}

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

// per-lane constant table of k_fast2048 (the rings-in-LDS kernel), entry-major: tab[e*64 + lane]
constexpr int kTabWin = 0;         // 32 floats: w[2(l+64j)], w[2(l+64j)+1]
constexpr int kTabTw1 = 32;        // 15 complex: W1024^(l*k2), k2 = 1..15
constexpr int kTabTw2 = 62;        // 7 complex:  W64^(a*d),   d = 1..7, a = l&7
constexpr int kTabTwp = 76;        // 9 complex:  W2048^(kP[u]), u = 0..8
constexpr int kTabEntries = 94;

// per-lane constant table of the register-FFT kernels
constexpr int kT2Win = 0;       // 16: (w[2(l+64j)], w[2(l+64j)+1])
constexpr int kT2Tw1 = 16;      // 15: W1024^(l*k2), k2 = 1..15     (every group starts on an even entry: pair reads)
constexpr int kT2Tw2 = 32;      // 7:  W64^(a*d), d = 1..7
constexpr int kT2Twp = 40;      // 9:  -i * W2048^(kP[u])
constexpr int kT2Entries = 50;  // entries the n_fft-2048 kernels copy to LDS (25 KB)
constexpr int kT2Zero = 50;     // global-only: one row of zeros after the table (the "no tail" operand of load_row)
// Entries are stored in pairs, [e/2][lane][e&1]: a lane's entries 2p and 2p+1 are 16 contiguous bytes, so
// constants that are used together (two window values, two twiddles) arrive with one ds_read_b128
// (full LDS rate) instead of two reads.  tabl = smem + 2*lane; entry e of the lane is tabl[TE(e)].
#define TE(e) ((((e) >> 1) * 128) + ((e) & 1))
// Where the per-lane constants come from: the LDS copy of the table (entry e of the lane at tabl[TE(e)]) ...
struct TabLds {
    const v2f* tabl;
    __device__ __forceinline__ v2f get(int e) const { return lds_rd(&tabl[TE(e)]); }
    __device__ __forceinline__ void pair(int e, v2f& a, v2f& b) const { lds_rd_pair(&tabl[TE(e)], a, b); }
};
// ... or, for the kernels that have the registers (2 waves per SIMD), the two FFT twiddle groups (entries kT2Tw1 .. kT2Twp-1) from registers, windows and split twiddles from LDS
struct TabMixed {
    const v2f* tabl;
    const v2f* r;        // r[e - kT2Tw1]
    __device__ __forceinline__ bool in_reg(int e) const { return e >= kT2Tw1 && e < kT2Twp; }
    __device__ __forceinline__ v2f get(int e) const {
#if defined(GL_EXPERIMENT) && defined(GL_ABL)
        if ((GL_ABL & 2) && !in_reg(e)) return r[e & 15];      // timing study: no table reads
#endif
        return in_reg(e) ? r[e - kT2Tw1] : lds_rd(&tabl[TE(e)]);
    }
    __device__ __forceinline__ void pair(int e, v2f& a, v2f& b) const {
#if defined(GL_EXPERIMENT) && defined(GL_ABL)
        if ((GL_ABL & 2) && !in_reg(e)) { a = r[e & 15]; b = r[(e + 1) & 15]; return; }
#endif
        if (in_reg(e)) { a = r[e - kT2Tw1]; b = r[e + 1 - kT2Tw1]; }
        else lds_rd_pair(&tabl[TE(e)], a, b);
    }
};
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
__device__ __forceinline__ void twiddle2(v2f* z, const TAB& tb) {   // z[d], z[8+d] *= W64^(+-a*d)
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {       // d = 2pp+1, 2pp+2
        v2f wa, wb;
        if (pp < 3) tb.pair(kT2Tw2 + 2 * pp, wa, wb);
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
#if defined(GL_EXPERIMENT) && defined(GL_ABL)
// Timing study (GL_ABL & 32; results are NOT a transform): the cost of exchange 1 -- register bits 0..2 against lane
// bits 3..5 -- done in the register file instead of LDS, as the three butterfly stages of an 8 x 8 transpose:
// v_permlane32_swap (lane bit 5), v_permlane16_swap (lane bit 4), each one instruction per VGPR pair, and lane
// bit 3 by two DPP row_ror:8 selects per VGPR pair.  16 + 16 + 32 = 64 VALU operations per exchange.
__device__ __forceinline__ void ex1_in_registers(v2f* z, int l) {
    const bool hi = (l & 8) != 0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {          // register bit 0 <-> lane bit 5
        auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(z[k].x), __float_as_uint(z[k + 1].x), false, false);
        auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(z[k].y), __float_as_uint(z[k + 1].y), false, false);
        z[k].x = __uint_as_float(a[0]); z[k + 1].x = __uint_as_float(a[1]);
        z[k].y = __uint_as_float(b[0]); z[k + 1].y = __uint_as_float(b[1]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {             // register bit 1 <-> lane bit 4
        if (k & 2) continue;
        auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(z[k].x), __float_as_uint(z[k + 2].x), false, false);
        auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(z[k].y), __float_as_uint(z[k + 2].y), false, false);
        z[k].x = __uint_as_float(a[0]); z[k + 2].x = __uint_as_float(a[1]);
        z[k].y = __uint_as_float(b[0]); z[k + 2].y = __uint_as_float(b[1]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {             // register bit 2 <-> lane bit 3: rotate by 8 inside each row of 16 lanes
        if (k & 4) continue;
        const float ax = z[k].x, ay = z[k].y, bx = z[k + 4].x, by = z[k + 4].y;
        const float rbx = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(bx), 0x128, 0xf, 0xf, false));
        const float rby = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(by), 0x128, 0xf, 0xf, false));
        const float rax = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(ax), 0x128, 0xf, 0xf, false));
        const float ray = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(ay), 0x128, 0xf, 0xf, false));
        z[k].x = hi ? rbx : ax;      z[k].y = hi ? rby : ay;
        z[k + 4].x = hi ? bx : rax;  z[k + 4].y = hi ? by : ray;
    }
}
#endif

#if defined(GL_EXPERIMENT) && defined(GL_ABL)
#define GL_ABL_SKIP_EX if (GL_ABL & 1) return
#define GL_ABL_EX1_REG_PUT if (GL_ABL & 32) return
#define GL_ABL_EX1_REG_GET if (GL_ABL & 32) { ex1_in_registers(z, lp.a3 + 8 * lp.g3); return; }
#else
#define GL_ABL_SKIP_EX
#define GL_ABL_EX1_REG_PUT
#define GL_ABL_EX1_REG_GET
#endif
__device__ __forceinline__ void ex1_put_cols(const v2f* z, v2f* ex, const LanePos& lp) {
    GL_ABL_SKIP_EX;
    GL_ABL_EX1_REG_PUT;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) ex[k2 * kE1S + lp.piL] = z[k2];
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void ex1_get_rows(v2f* z, const v2f* ex, const LanePos& lp) {
    GL_ABL_SKIP_EX;
    GL_ABL_EX1_REG_GET;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int bb = 0; bb < 8; bb += 2)
            lds_rd_pair(&ex[(lp.g3 + 8 * h) * kE1S + lp.a3 * kE1P + bb], z[8 * h + bb], z[8 * h + bb + 1]);
}
__device__ __forceinline__ void ex1_put_rows(const v2f* z, v2f* ex, const LanePos& lp) {
    GL_ABL_SKIP_EX;
    GL_ABL_EX1_REG_PUT;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int bb = 0; bb < 8; bb += 2)
            lds_wr_pair(&ex[(lp.g3 + 8 * h) * kE1S + lp.a3 * kE1P + bb], z[8 * h + bb], z[8 * h + bb + 1]);
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void ex1_get_cols(v2f* z, const v2f* ex, const LanePos& lp) {
    GL_ABL_SKIP_EX;
    GL_ABL_EX1_REG_GET;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) z[k2] = lds_rd(&ex[k2 * kE1S + lp.piL]);
}
// exchange 2: register d <-> lane bits a3.  "rows" side: 2 x 4 x b128; "cols" side: 16 x b64 at rho(l), rho(128-l)
__device__ __forceinline__ void ex2_put_rows(const v2f* z, v2f* ex, const LanePos& lp) {
    GL_ABL_SKIP_EX;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int d = 0; d < 8; d += 2)
            lds_wr_pair(&ex[lp.a3 * kE2S + (lp.g3 + 8 * h) * kE2Q + d], z[8 * h + d], z[8 * h + d + 1]);
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void ex2_get_cols(v2f* z, const v2f* ex, const LanePos& lp) {
    GL_ABL_SKIP_EX;
#pragma unroll
    for (int a = 0; a < 8; ++a) { z[a] = lds_rd(&ex[a * kE2S + lp.rhoL]); z[8 + a] = lds_rd(&ex[a * kE2S + lp.rhoK]); }
}
__device__ __forceinline__ void ex2_put_cols(const v2f* z, v2f* ex, const LanePos& lp) {
    GL_ABL_SKIP_EX;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int a = 0; a < 8; ++a) { ex[a * kE2S + lp.rhoL] = z[a]; ex[a * kE2S + lp.rhoK] = z[8 + a]; }
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void ex2_get_rows(v2f* z, const v2f* ex, const LanePos& lp) {
    GL_ABL_SKIP_EX;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int d = 0; d < 8; d += 2)
            lds_rd_pair(&ex[lp.a3 * kE2S + (lp.g3 + 8 * h) * kE2Q + d], z[8 * h + d], z[8 * h + d + 1]);
}

template <class TAB>
__device__ __forceinline__ void fft_forward_1024(v2f* z, v2f* ex, int l, int kloB, const TAB& tb, Stamps& st) {
    const LanePos lp(l, kloB);
    dft16<false>(z);
    twiddle1<false>(z, tb);
    st.mark(1);
    ex1_put_cols(z, ex, lp);
    ex1_get_rows(z, ex, lp);
    st.mark(2);
    dft8<false>(z);
    dft8<false>(z + 8);
    twiddle2<false>(z, tb);
    st.mark(3);
    ex2_put_rows(z, ex, lp);
    ex2_get_cols(z, ex, lp);
    st.mark(4);
    dft8<false>(z);
    dft8<false>(z + 8);
    st.mark(5);
}

template <class TAB>
__device__ __forceinline__ void fft_inverse_1024(v2f* z, v2f* ex, int l, int kloB, const TAB& tb, Stamps& st) {
    const LanePos lp(l, kloB);
    dft8<true>(z);
    dft8<true>(z + 8);
    st.mark(7);
    ex2_put_cols(z, ex, lp);
    ex2_get_rows(z, ex, lp);
    st.mark(8);
    twiddle2<true>(z, tb);
    dft8<true>(z);
    dft8<true>(z + 8);
    st.mark(9);
    ex1_put_rows(z, ex, lp);
    ex1_get_cols(z, ex, lp);
    st.mark(10);
    twiddle1<true>(z, tb);
    dft16<true>(z);
    st.mark(11);
}

// Lane 0 owns the two self-mirrored butterflies (k_lo = 0 and 64), so its nine units pair registers differently
// from every other lane's.  Instead of 24 two-instruction selects per frame, the twelve registers that differ
// are rotated for lane 0 alone: twelve v_mov_b64 under EXEC = lane 0, in an order that needs no temporary
// (the wave runs with all 64 lanes enabled: the kernels have no divergent control flow around this point).
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
__device__ __forceinline__ void middle_project(v2f* z, const float* m_lo, const float* m_hi, const TAB& tb) {
    v2f P[9], Q[9];
    const v2f z4 = z[4];
    // lane 0: P[4..7] = z[8..11], Q[0] = z[0], Q[1..3] = z[7..5], Q[4..7] = z[15..12]; with P[c] living in z[c]
    // and Q[c] in z[15-c] that is the rotation z4<-z8<-z12<-z5<-z9<-z13<-z6<-z10<-z14<-z7<-z11<-z15<-z0
    lane0_chain(z[4], z[8], z[12], z[5], z[9], z[13], z[6], z[10], z[14], z[7], z[11], z[15], z[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) { P[c] = z[c]; Q[c] = z[15 - c]; }
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
    // results live in R_c = P[c] (-> z[c]) and S_i = Q[7-i] (-> z[8+i]); lane 0 wants z4 = P8, z5..7 = Q3..1,
    // z8..11 = P4..7, z12..15 = Q7..4: the rotation S7<-S3<-R7<-S6<-S2<-R6<-S5<-S1<-R5<-S4<-S0<-R4<-P8
    lane0_chain(Q[0], Q[4], P[7], Q[1], Q[5], P[6], Q[2], Q[6], P[5], Q[3], Q[7], P[4], P[8]);
#pragma unroll
    for (int c = 0; c < 8; ++c) { z[c] = P[c]; z[8 + c] = Q[7 - c]; }
}

// The same step for the n_fft-2048 kernels: magnitudes arrive as pairs mm[u] = (m[k_u], m[1024 - k_u]) (packed
// layout below) and the zero-frame rule is the planted kZeroNudge (project2), so no per-bin nudge or clamp.
template <class TAB>
__device__ __forceinline__ void middle_project2(v2f* z, const v2f* mm, float m512, const TAB& tb) {
    v2f P[9], Q[9];
    const v2f z4 = z[4];
    lane0_chain(z[4], z[8], z[12], z[5], z[9], z[13], z[6], z[10], z[14], z[7], z[11], z[15], z[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) { P[c] = z[c]; Q[c] = z[15 - c]; }
    auto unit = [&](int u, v2f w) {
        const v2f uu = pk_addc(P[u], Q[u]), vv = pk_subc(P[u], Q[u]);
        v2f X = cmul_add(vv, w, uu), Y = cmul_sub(vv, w, uu);
        project2(X, Y, mm[u]);
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
    {
        const float r = fmaf(z4.x, z4.x, fmaf(z4.y, z4.y, 0x1p-126f));
        const float sc = (m512 + m512) * __builtin_amdgcn_rsqf(r);
        P[8] = z4 * (v2f){sc, sc};
    }
    lane0_chain(Q[0], Q[4], P[7], Q[1], Q[5], P[6], Q[2], Q[6], P[5], Q[3], Q[7], P[4], P[8]);
#pragma unroll
    for (int c = 0; c < 8; ++c) { z[c] = P[c]; z[8 + c] = Q[7 - c]; }
}

// Magnitudes as the register-ring kernel's lanes consume them.  Lane l of a frame needs the 17
// values m[k], m[1024-k] at its 8 mirror pairs k = l + 128u (lane 0: see middle_project) and
// m[512]: 17 dword loads per frame, rows 4100 bytes long so every load straddles three 128-byte
// lines.  The kernel is bound by the number of line requests a CU can keep in flight (TCP
// pending-stall 46 % of cycles), so gl_run first repacks the spectrogram once into
// [frame][quad q][lane][4] + [frame][1024 + lane]: four aligned dwordx4 loads and one dword per
// frame, 34 line requests instead of about 51.  4352 bytes per frame (6 % more than 4100).  Quad q of a lane
// holds (m[k], m[1024-k]) of its mirror pairs u = 2q and 2q+1, so a pair's two magnitudes are one register pair.
constexpr int kMagPack = 1088;
// 1/M of the inverse transform and the factor 2 the split step leaves, folded into the magnitudes once
// (a power of two: exact), so the forward and the inverse window are the same table entries
constexpr float kMagScale = 1.0f / (2.0f * kM);
typedef float v4f __attribute__((ext_vector_type(4)));

}  // namespace gl_fft
