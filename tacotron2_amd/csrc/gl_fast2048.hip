// Fused wave-per-frame Griffin-Lim kernels for gfx950 (MI355X), n_fft = 2048.
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
//   k_fast2048<MODE>     any hop; rings in LDS; also the single-shot STFT / ISTFT
//   k_gl2048_regring<HS> hop 128*HS: rings in registers, two frames in flight per wave (the benchmark kernel)
//   k_gl2048_frames<HS>  hop 256*k, one or two utterances: one frame per wave, state kept as frames
//   k_analysis2048       |centred STFT| of the forward analysis path (wav -> log-mel), one wave per frame
// The FFT core (dft16/dft8, twiddle1/2, ex1_*/ex2_* transposes, middle_project) is gl_fft1024.h.
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
#include <string.h>

#include <vector>

#include "gl_fast_host.h"
#include "gl_fft1024.h"

using namespace gl_fft;

namespace {

constexpr int kRing = kN;          // ring length (floats), power of two
constexpr int kEx1Stride = 72;     // exchange-1 row stride (complex), 64 + 8 pad
constexpr int kEx2Stride = 130;    // exchange-2 row stride (complex), 128 + 2 pad
constexpr int kScratch = 16 * kEx1Stride;   // complex elements (>= 8*kEx2Stride)
static_assert(kScratch >= 8 * kEx2Stride, "scratch too small");

enum { MODE_GL = 0, MODE_STFT = 1, MODE_ISTFT = 2 };

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
// window / split-twiddle constants (25 KB table shared by the workgroup); the FFT twiddles
// live in registers.  8 waves per CU (2 per SIMD, 242 VGPRs); the frame loop keeps two
// frames in flight per wave (see the loop).
// =====================================================================================
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
    for (int q = 0; q < 4; ++q)     // (m[k], m[1024-k]) of the mirror pairs u = 2q, 2q+1
        *reinterpret_cast<v4f*>(orow + (q * 64 + l) * 4) = (v4f){v[2 * q], v[8 + 2 * q], v[2 * q + 1], v[8 + 2 * q + 1]};
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

// 8 waves in one workgroup per CU = 2 waves per SIMD with up to 256 VGPRs each.  The 22 FFT twiddles of a lane
// live in registers, windows and split twiddles are read from the LDS copy of the table at their use -- what fits
// beside the second frame of the pipelined loop.  Measured on config 2 (ms per iteration): 12 waves + everything
// from LDS 0.149, 8 waves + LDS 0.154, 8 waves + all 47 constants in registers 0.1435 (unpipelined) / spills
// (pipelined), pipelined + twiddles in registers 0.1420.
constexpr int kRRWaves = 8;        // waves per workgroup; they only share the constant table

template <int HS>
__global__ void __launch_bounds__(64 * kRRWaves, 1) k_gl2048_regring(AlignedArgs p) {
    // one LDS array (table + per-wave transpose scratch): 25 KB + 8 x 10.4 KB = 108 KB -> 1 WG / CU
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
    const int kloB = l ? 128 - l : 64;
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
    // magnitudes at the lane's 9 mirror pairs of frame `fr` (packed layout, see k_mag_repack): mm[u] = (m[k], m[1024-k])
    auto load_mags = [&](int fr, v2f* mm, float& m512) {
#if defined(GL_EXPERIMENT) && defined(GL_ABL)
        if (GL_ABL & 4) {                      // timing study: no magnitude loads
#pragma unroll
            for (int u = 0; u < 8; ++u) mm[u] = (v2f){1.0f + u, 2.0f + u};
            m512 = 3.0f;
            return;
        }
#endif
#if defined(GL_EXPERIMENT) && defined(GL_ABL)
        if (GL_ABL & 16) fr = f0 + (fr & 1);   // timing study: the same two frames' magnitudes over and over (cache hits)
#endif
        const float* mrow = p.magp + ((long long)b * T + fr) * kMagPack + 4 * l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // streamed once per launch: nontemporal, so the 240 MB do not push x and the tails out of L2 (-1.3 %)
            const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(mrow + 256 * q));
            mm[2 * q] = t.xy;
            mm[2 * q + 1] = t.zw;
        }
        m512 = mrow[1024 - 3 * l];
    };
    v2f mm[8];
    float m512;
    load_mags(f0, mm, m512);
    for (int i = threadIdx.x; i < kT2Entries * 64; i += 64 * kRRWaves) smem[i] = p.tab[i];
    __syncthreads();
    if (!active) return;                         // wave-uniform; no barrier after this point
    v2f* ex = smem + kT2Entries * 64 + wave * kRScratch;
    const v2f* tabl = smem + 2 * l;
    v2f treg[kT2Twp - kT2Tw1];
#pragma unroll
    for (int e = kT2Tw1; e < kT2Twp; ++e) treg[e - kT2Tw1] = lds_rd(&tabl[TE(e)]);
    const TabMixed tb{tabl, treg};
    // sample 0 of every windowed frame is 0 (w[0] = 0): lane 0 plants kZeroNudge there instead (see project2)
    const v2f nudge = {l == 0 ? kZeroNudge : 0.0f, 0.0f};

    // Drain the ring-fill loads here.  Otherwise the loop header inherits "xraw[8..15] still in
    // flight" from this prologue and the compiler guards every frame's window multiply with
    // s_waitcnt vmcnt(24..17), which in steady state waits for the rows it has just requested
    // for the NEXT frame: one exposed global-load round trip per frame.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
    Stamps st;
    st.start();

    // Software pipeline over two frames.  Frames of one iteration are independent (every frame reads x_in
    // and adds into x_out), so the forward transform of frame n+1 ("A") is interleaved with the projection
    // and inverse transform of frame n ("B"): each of the four LDS transposes is issued and then followed
    // by arithmetic of the OTHER frame, so its round trip is covered by work of the same wave instead of
    // depending on what the one other wave of the SIMD happens to be doing.  Both frames use the same
    // scratch: LDS operations of a wave execute in program order, and the order below alternates
    // A.ex1, B.ex2, A.ex2, B.ex1, each written and read completely before the next is written.
    // (Scheduling fences between the phases, so that the consumers of a transpose's reads cannot be hoisted
    // in front of the other frame's arithmetic, were measured: +1.5 % -- the scheduler's own order is better.)
    // A step ends with B's time-domain frame: its overlap-add, the hand-over of A's spectrum into B's registers
    // and the windowing of frame n+2 into A's share one pass over the window table.  (Unrolling by two so that the
    // arrays swap roles instead of being copied spills: 256 VGPRs + 100 bytes of scratch.)
    const LanePos lp(l, kloB);
    auto window_in = [&](v2f* z) {
        z[0] = __builtin_elementwise_fma(xraw[0], tb.get(kT2Win), nudge);
#pragma unroll
        for (int j = 1; j < 16; ++j) z[j] = xraw[j] * tb.get(kT2Win + j);
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
    // the HS finished rows of frame n leave; the accumulator slides
#if defined(GL_EXPERIMENT) && defined(GL_PAR)
    // parity attribution build GL_PAR & 4: overlap-add in two-float arithmetic (exact product by fma, TwoSum into a
    // low-order accumulator that joins the sum when the row leaves); spills, timing is meaningless
    v2f accl[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) accl[j] = (v2f){0.0f, 0.0f};
    auto ola_add = [&](int j, v2f z, v2f w) {
        if (GL_PAR & 4) {
            const v2f pr = z * w;
            const v2f pe = __builtin_elementwise_fma(z, w, -pr);
            const v2f sum = accr[j] + pr;
            const v2f bb = sum - accr[j];
            const v2f err = (accr[j] - (sum - bb)) + (pr - bb);
            accr[j] = sum;
            accl[j] += err + pe;
        } else {
            accr[j] = __builtin_elementwise_fma(z, w, accr[j]);
        }
    };
#else
    auto ola_add = [&](int j, v2f z, v2f w) { accr[j] = __builtin_elementwise_fma(z, w, accr[j]); };
#endif
    auto store_rows = [&](int n) {
        float* orow = p.x_out + (long long)b * p.L + (long long)n * H + 2 * l;
#if defined(GL_EXPERIMENT) && defined(GL_PAR)
        if (GL_PAR & 4) {
#pragma unroll
            for (int jj = 0; jj < HS; ++jj) accr[jj] += accl[jj];
#pragma unroll
            for (int j = 0; j < OVR; ++j) accl[j] = accl[j + HS];
#pragma unroll
            for (int jj = 0; jj < HS; ++jj) accl[OVR + jj] = (v2f){0.0f, 0.0f};
        }
#endif
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) *reinterpret_cast<v2f*>(orow + 128 * jj) = accr[jj];
#pragma unroll
        for (int j = 0; j < OVR; ++j) accr[j] = accr[j + HS];
#pragma unroll
        for (int jj = 0; jj < HS; ++jj) accr[OVR + jj] = (v2f){0.0f, 0.0f};
    };

#if defined(GL_EXPERIMENT) && defined(GL_REPS)
    // timing study (wrong numbers): GL_REPS iterations inside ONE launch with NO hand-over between runs -- the upper
    // bound of what a persistent several-iterations-per-launch kernel with neighbour flags could gain (it would
    // pay for its flags and cache maintenance on top).  The host launches iters / GL_REPS kernels.
#pragma nounroll
    for (int rep = 0; rep < GL_REPS; ++rep) {
    if (rep) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { xraw[j] = load_row(128 * j); accr[j] = (v2f){0.0f, 0.0f}; }
        load_mags(f0, mm, m512);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
#endif
    // prologue: forward transform of the run's first frame (zB), windowed input of its second (zA)
    v2f zA[16], zB[16];
    {
        v2f xn[HS], xt[HS];
        rows_request(f0 + 1, f0 + 1 < f1, xn, xt);
        window_in(zB);
        fft_forward_1024(zB, ex, l, kloB, tb, st);
        ring_advance(xn, xt);                        // xraw = input of frame f0+1
        window_in(zA);
    }
    // zA: windowed input of frame n+1 -> its spectrum; zB: spectrum of frame n -> its inverse transform
    for (int n = f0; n + 1 < f1; ++n) {
        v2f xn[HS], xt[HS];
        rows_request(n + 2, n + 2 < f1, xn, xt);
        dft16<false>(zA);                            // A
        twiddle1<false>(zA, tb);
        ex1_put_cols(zA, ex, lp);
        ex1_get_rows(zA, ex, lp);
        middle_project2(zB, mm, m512, tb);           // B
        load_mags(n + 1, mm, m512);
        dft8<true>(zB);
        dft8<true>(zB + 8);
        ex2_put_cols(zB, ex, lp);
        ex2_get_rows(zB, ex, lp);
        dft8<false>(zA);                             // A
        dft8<false>(zA + 8);
        twiddle2<false>(zA, tb);
        ex2_put_rows(zA, ex, lp);
        ex2_get_cols(zA, ex, lp);
        twiddle2<true>(zB, tb);                      // B
        dft8<true>(zB);
        dft8<true>(zB + 8);
        ex1_put_rows(zB, ex, lp);
        ex1_get_cols(zB, ex, lp);
        dft8<false>(zA);                             // A: spectrum of frame n+1 complete
        dft8<false>(zA + 8);
        twiddle1<true>(zB, tb);                      // B
        dft16<true>(zB);
        ring_advance(xn, xt);                        // xraw = input of frame n+2
        // zB = inverse transform of frame n: window it into the accumulator; zB then takes over zA's spectrum
        // (frame n+1, the next step's B) and zA becomes the windowed input the ring holds (frame n+2, the next
        // step's A): one pass over the window table for all three (1/2048 is folded into the packed magnitudes,
        // so the analysis and the synthesis window are the same entries)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const v2f w = tb.get(kT2Win + j);
            ola_add(j, zB[j], w);
            zB[j] = zA[j];
            zA[j] = (j == 0) ? __builtin_elementwise_fma(xraw[0], w, nudge) : xraw[j] * w;
        }
        store_rows(n);
    }
    {   // epilogue: the run's last frame has no successor to overlap with
        middle_project2(zB, mm, m512, tb);
        fft_inverse_1024(zB, ex, l, kloB, tb, st);
#pragma unroll
        for (int j = 0; j < 16; ++j) ola_add(j, zB[j], tb.get(kT2Win + j));
        store_rows(f1 - 1);
    }
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
#if defined(GL_EXPERIMENT) && defined(GL_REPS)
    }
#endif
}

// =====================================================================================
// Small problems at n_fft 2048, hop 128*HS -- k_gl2048_frames<HS> + k_ola_frames.  Same idea as k_gl1024_frames
// (gl_fast1024.hip): when B*T frames cannot fill the GPU with runs (one utterance; one rank's chunk of a long utterance), a
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
    v2f mm[8];
    float m512;
    {
        const float* mrow = p.magp + ((long long)b * T + n) * kMagPack + 4 * l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4f t = *reinterpret_cast<const v4f*>(mrow + 256 * q);
            mm[2 * q] = t.xy;
            mm[2 * q + 1] = t.zw;
        }
        m512 = mrow[1024 - 3 * l];
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
    const int kloB = l ? 128 - l : 64;

#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = z[j] * tb.get(kT2Win + j);
    if (l == 0) z[0].x = kZeroNudge;             // sample 0 of the windowed frame is 0 (w[0] = 0): see project2
    Stamps st;
    fft_forward_1024(z, ex, l, kloB, tb, st);
    middle_project2(z, mm, m512, tb);
    fft_inverse_1024(z, ex, l, kloB, tb, st);
    float* yo = p.y_out + ((long long)b * T + n) * kN + 2 * l;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const v2f w = tb.get(kT2Win + j);                             // 1/2048 is folded into the packed magnitudes
        *reinterpret_cast<v2f*>(yo + 128 * j) = z[j] * w;
    }
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
        fft_forward_1024(z, ex, l, kloB, tb, st);
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

// the register-ring kernels exist for hops of 1, 2, 4 or 8 register rows and need 8-byte aligned signals
bool regring_ok(const gl_plan* p, const void* a, const void* b) {
    return gl_fast_regring_hop(p) && (((uintptr_t)a | (uintptr_t)b) & 7) == 0;
}

int launch_regring(const gl_plan* p, const AlignedArgs& a, int B, hipStream_t st) {
    dim3 grid((unsigned)(((long long)a.nruns * B + kRRWaves - 1) / kRRWaves));
    dim3 block(64 * kRRWaves);
    switch (p->hop / 128) {
        case 1: hipLaunchKernelGGL(k_gl2048_regring<1>, grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL(k_gl2048_regring<2>, grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL(k_gl2048_regring<4>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(k_gl2048_regring<8>, grid, block, 0, st, a); break;
    }
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int run_regring(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                hipStream_t st) {
    const int G = gl_fast_pick_run_length(p, B, T, kRRWaves);
    const int nruns = (T + G - 1) / G;
    if (mag != nullptr) {            // null: the mel inversion stored the packed layout itself (gl_mel_run)
        hipLaunchKernelGGL(k_mag_repack, dim3(T, B), dim3(64), 0, st, mag, p->d_magp, T);
        GL_HIP_CHECK(hipGetLastError());
    }
    // Too few frames to fill the GPU with runs: one wave per frame, state kept as frames (k_gl2048_frames).  Not
    // with a pinned run length (callers who pin it compare bits with the run kernel) nor with the optional
    // per-iteration signal transforms.  Measured at hop 256, T = 861 (us per iteration, per-frame form vs
    // register-ring runs): B=1 22.0 / 34.6, B=2 23.4 / 35.0, B=4 45.5 / 35.9, one 6 600-frame chunk 89.7 / 37.1:
    // the run kernel scales far better (no 8-fold gather), hence gl_fast_small_form_limit = 8 frames per CU.
    if (!gl_post_needed(p) && p->run_frames == 0 && (long long)B * T <= gl_fast_small_form_limit(p)) {
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
        return gl_fast_ola_frames(ybuf[(iters - 1) & 1], x_out, B, T, kN, p->hop, st);
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
    int rc = 0;
#if defined(GL_EXPERIMENT) && defined(GL_REPS)
    iters = (iters + GL_REPS - 1) / GL_REPS;
#endif
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
#if defined(GL_EXPERIMENT) && defined(GL_STAMPS)
    {   // kernel studies only (abl/stamps.sh): per-phase shader-clock totals of the Stamps instrumentation
        unsigned long long h[24];
        GL_HIP_CHECK(hipStreamSynchronize(st));
        GL_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h)));
        const double frames = (double)B * T * iters;
        double tot = 0;
        for (int i = 0; i < 16; ++i) tot += (double)h[i];
        fprintf(stderr, "[stamps] shader-clock ticks per frame per wave (G=%d, waves %llu): total %.0f\n", G, h[20], tot / frames);
        for (int i = 0; i < 16; ++i) fprintf(stderr, "  phase %2d: %8.1f\n", i, (double)h[i] / frames);
        memset(h, 0, sizeof(h));
        GL_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), h, sizeof(h)));
    }
#endif
    if (nruns > 1 && !post) return gl_fast_fold(p, src, src_tail, x_out, B, T, G, nruns, st);
    return 0;
}

}  // namespace

int gl_fast_stft(gl_plan* p, const float* x, long long x_stride, float* S, int B, int T, hipStream_t st) {
    FastArgs a = {};
    a.x_in = x;
    a.x_stride = x_stride;
    a.S_out = (float2*)S;
    a.tab = p->d_fast_tab;
    a.H = p->hop;
    a.T = T;
    a.G = gl_fast_pick_run_length(p, B, T, 4);
    a.nruns = (T + a.G - 1) / a.G;
    a.L = gl_len(p->n_fft, p->hop, T);
    return launch<MODE_STFT>(a, B, st);
}

int gl_fast_istft(gl_plan* p, const float* S, float* y, int B, int T, hipStream_t st) {
    FastArgs a = {};
    a.S_in = (const float2*)S;
    a.tab = p->d_fast_tab;
    a.H = p->hop;
    a.T = T;
    a.G = gl_fast_pick_run_length(p, B, T, 4);
    a.nruns = (T + a.G - 1) / a.G;
    a.L = gl_len(p->n_fft, p->hop, T);
    a.x_out = (a.nruns > 1) ? p->d_xa : y;
    a.tail_out = p->d_tail_a;
    int rc = launch<MODE_ISTFT>(a, B, st);
    if (rc || a.nruns == 1) return rc;
    return gl_fast_fold(p, p->d_xa, p->d_tail_a, y, B, T, a.G, a.nruns, st);
}

// N3: |centred STFT| of B signals (n_fft 2048 only; gl_analysis_mag is the any-N path)
int gl_fast_analysis_mag(gl_plan* p, const float* y, long long y_stride, long long len, float* mag, int reflect,
                         int B, int T, const float* win_p, hipStream_t st) {
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

bool gl_fast_takes_packed(const gl_plan* p, const void* x0, const void* x_out) {
    return p->path == 0 && p->n_fft == 2048 && p->d_magp != nullptr && regring_ok(p, x0, x_out);
}
int gl_fast_run_packed(gl_plan* p, const float* x0, float* x_out, int iters, int B, int T, hipStream_t st) {
    return run_regring(p, nullptr, x0, x_out, iters, B, T, st);
}

// the Griffin-Lim loop at n_fft 2048: register rings where the hop allows, else the rings-in-LDS kernel
int gl_fast2048_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                    hipStream_t st) {
    if (regring_ok(p, x0, x_out)) return run_regring(p, mag, x0, x_out, iters, B, T, st);
    const int G = gl_fast_pick_run_length(p, B, T, 4);
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
    int rc = 0;
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
    if (nruns > 1 && !post) return gl_fast_fold(p, src, src_tail, x_out, B, T, G, nruns, st);
    return 0;
}
