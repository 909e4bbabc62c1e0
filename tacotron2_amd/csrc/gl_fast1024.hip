// Fused wave-per-frame Griffin-Lim kernels for gfx950 (MI355X), n_fft = 1024, any hop (the reference's
// params.py defaults are 1024 / 275).  FFT core shared with the 2048 kernels (gl_fft1024.h).
#include <math.h>
#include <string.h>

#include <type_traits>

#include "gl_fast_host.h"
#include "gl_fft1024.h"

using namespace gl_fft;

namespace {

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
// The overlap-add accumulator lives in a per-wave LDS ring (hop 275 is not a multiple of
// the 64-sample register rows).
// =====================================================================================
constexpr int kN1 = 1024;
constexpr int kP1Ring = 2048;      // accumulator ring, floats; >= N + H for every hop <= N

struct Pair1024Args {
    const float* x_in;      // [B][L]
    const float* mag;       // [B][T][513]
    float* x_out;           // [B][L]
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
__device__ __forceinline__ void middle_two_frames(v2f* z, const float* ma, const float* mb, float ka, float kb) {
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
}
// The run kernel: 8 waves per CU, a wave owns a run of frames of one utterance.  A frame's samples come straight
// from global memory (consecutive frames overlap, so they hit L1/L2) and 1/1024 is folded into the magnitudes so
// the inverse window is the forward window.  LDS: 8 x (8 KB accumulator ring + 10.4 KB transpose scratch) + 4 KB
// analysis window + 4 KB synthesis window (the same values times 1/1024, the inverse transform's scale: exact, and 18
// multiplications of the magnitudes fewer per pair) + 4 KB second-pass twiddles = 159 KB; the 15 first-pass twiddles
// of a lane live in registers.  Runs of all utterances are numbered consecutively, like the register-ring kernel's.
//
// Two pairs in flight per wave, as in k_gl2048_regring: pairs of one iteration are independent, so the forward
// transform of pair n+2 ("A") is interleaved phase by phase with the projection and inverse transform of pair n
// ("B"); every LDS transpose is followed by arithmetic of the other pair.  Raw samples are requested two pairs
// ahead (in the step that ends by windowing them), magnitudes one pair ahead.
//
// Overlap-add.  Frame position i = l + 64 j of frame m is signal sample s = m H + i, ring slot s mod 2048.  A
// sample is first touched by the frame that holds it among its LAST H positions (i >= N - H) and complete after the
// frame that holds it among its FIRST H (i < H).  With the hop known at compile time (HT = 275, the reference's
// params.py) each register j falls into one class -- or two, split at a fixed lane:
//     i >= N - H          ds_write only   (first touch: no clearing of old contents anywhere)
//     H <= i < N - H      ds_read, add, ds_write
//     i < H               ds_read, add, global store (the slot is simply overwritten 2048 samples later)
// 23 LDS operations per frame instead of 32 and no selects.  (ds_add_f32 for the middle class was measured: LDS float
// atomics take ~190 LDS cycles per wave instruction, the kernel ran twice as long.)  HT = 0 is the same kernel for
// any hop from the arguments: every slot read, added, written back (zero where complete).
constexpr int kP2Waves = 8;
constexpr int kP2RingBytes = kP1Ring * 4;                       // 8 KB, and every wave's ring is 8 KB aligned
constexpr int kP2ExBytes = kRScratch * 8;
struct Pair2Args {
    Pair1024Args a;
    int B;
    float* x_zero;          // [B][L] the buffer the next launch writes: its seam regions are cleared here (or null)
};
// per-lane constants of the transforms: first-pass twiddles from registers, second-pass twiddles from an LDS
// table laid out [pair][lane] so that two arrive per ds_read_b128
struct TabP2 {
    const v2f* r;           // r[e - kT2Tw1], e in [kT2Tw1, kT2Tw2)
    const v4f* t2;          // t2[64 * pp] = (entry kT2Tw2 + 2 pp, entry kT2Tw2 + 2 pp + 1) of this lane
    __device__ __forceinline__ v2f get(int e) const {
        if (e < kT2Tw2) return r[e - kT2Tw1];
        const v4f t = t2[64 * ((e - kT2Tw2) >> 1)];
        return ((e - kT2Tw2) & 1) ? t.zw : t.xy;
    }
    __device__ __forceinline__ void pair(int e, v2f& a, v2f& b) const {
        if (e < kT2Tw2) { a = r[e - kT2Tw1]; b = r[e + 1 - kT2Tw1]; return; }
        const v4f t = t2[64 * ((e - kT2Tw2) >> 1)];
        a = t.xy;
        b = t.zw;
    }
};
template <int HT>
__global__ void __launch_bounds__(64 * kP2Waves, 1) k_gl1024_pair2(Pair2Args pp) {
    const Pair1024Args& p = pp.a;
    __shared__ __attribute__((aligned(8192))) unsigned char lds[kP2Waves * (kP2RingBytes + kP2ExBytes) + 3 * 4096];
    const int l = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // The run's first global loads (samples and magnitudes of its first pair, twiddles) are issued BEFORE the
    // workgroup's tables are built and the barrier, so the two round trips overlap; an idle wave of the last
    // workgroup goes through the motions on run 0 (which has no seam to clear) and leaves after the barrier.
    const int grun_raw = blockIdx.x * kP2Waves + wave;
    const bool active = grun_raw < pp.B * p.nruns;
    const int grun = active ? grun_raw : 0;
    // window: both frames of a pair see the same value, so a lane's 16 are scalars; stored [j / 4][lane][4] so that
    // four arrive per ds_read_b128 (in registers they cost 16 of the 256)
    v4f* wlds = reinterpret_cast<v4f*>(lds + kP2Waves * (kP2RingBytes + kP2ExBytes));
    v4f* t2lds = wlds + 4 * 64;
    v4f* wslds = wlds + 8 * 64;
    const int b = grun / p.nruns;
    const int run = grun - b * p.nruns;
    unsigned char* ring = lds + wave * kP2RingBytes;
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned ring_u = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;   // 8 KB aligned
    unsigned ring_v = ring_u;                                   // a VGPR copy: v_and_or_b32 takes one scalar operand
    asm volatile("" : "+v"(ring_v));
    float* acc = reinterpret_cast<float*>(ring);
    v2f* ex = reinterpret_cast<v2f*>(lds + kP2Waves * kP2RingBytes + wave * kP2ExBytes);
    v2f treg[kT2Tw2 - kT2Tw1];
    {
        const v2f* tg = p.tab + 2 * l;
#pragma unroll
        for (int e = kT2Tw1; e < kT2Tw2; ++e) treg[e - kT2Tw1] = tg[TE(e)];
    }
    const TabP2 tb{treg, t2lds + l};
    auto window4 = [&](int g) -> v4f { return wlds[g * 64 + l]; };
    auto swindow4 = [&](int g) -> v4f { return wslds[g * 64 + l]; };

    const int H = HT ? HT : p.H, T = p.T;
    const int f0 = run * p.G;
    const int f1 = min(f0 + p.G, T);
    const int ovl = kN1 - H;
    const int kloB = l ? 128 - l : 64;
    const LanePos lp(l, kloB);
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
    // lane part of the address in a VGPR, the constant part in the scalar offset operand: as one sum the compiler
    // turns 4 l + 256 j into an OR, hoists the 15 results out of the frame loop and spills them
    auto ldb = [](__amdgpu_buffer_rsrc_t rs, int lane_off, int const_off) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off, const_off, 0));
    };

#pragma unroll
    for (int i = 0; i < kP1Ring / 64; ++i) acc[l + 64 * i] = 0.0f;

    // raw samples of the pair starting at frame n (a lone last frame: b re-reads a and is discarded); buffer
    // loads: wave-uniform base in scalar registers + one 32-bit lane offset for all 32.  n is clamped into the
    // run by the callers, so the loads are unconditional.
    auto fetch_x = [&](int n, v2f (&xz)[16]) {
        const long long sn = (long long)n * H;
        const long long sb = (n + 1 < f1) ? sn + H : sn;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(xr + sn), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(xr + sb), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 16; ++j) { xz[j].x = ldb(rs_a, 4 * l, 256 * j); xz[j].y = ldb(rs_b, 4 * l, 256 * j); }
    };
    // magnitude index of the lane's 9 units, min(k, 1024-k), worked out at every use from a copy of the lane
    // number the compiler cannot see through: as loop invariants they would sit in 9 registers for the whole run
    auto fetch_mag = [&](int n, float (&fa)[9], float (&fb)[9]) {
        const float* ra = p.mag + ((long long)b * T + n) * (kN1 / 2 + 1);
        const float* rb = ra + ((n + 1 < f1) ? (kN1 / 2 + 1) : 0);
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)ra, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)rb, 0, 0x7fffffff, 0x00020000);
        int lv = l;
        asm volatile("" : "+v"(lv));
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int k = u == 8 ? 512 : (lv ? (lv + 128 * u) : (u < 4 ? 128 * u : 64 + 128 * (u - 4)));
            const int kk = k <= 512 ? k : 1024 - k;
            fa[u] = ldb(rs_a, 4 * kk, 0);
            fb[u] = ldb(rs_b, 4 * kk, 0);
        }
    };
    // windowed input of a pair from its raw samples.  An all-zero input frame must give an exactly zero spectrum
    // (phase 0 after projection, audio_utilities.py:181): its partner's rounding noise must not set its phase --
    // ka / kb = 0 then.  The test ORs the bit patterns (sign bits masked at the end) instead of comparing 32 values.
    auto window_pair = [&](bool two, const v2f (&xz)[16], v2f* z, float& ka, float& kb) {
        unsigned bits_a = 0, bits_b = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v4f w4;
            if ((j & 3) == 0) w4 = window4(j >> 2);
            bits_a |= __float_as_uint(xz[j].x);
            bits_b |= __float_as_uint(xz[j].y);
            z[j] = xz[j] * w4[j & 3];
        }
        if (!two) {
#pragma unroll
            for (int j = 0; j < 16; ++j) z[j].y = 0.0f;
            bits_b = 0;
        }
        ka = __any((bits_a & 0x7fffffffu) != 0) ? 1.0f : 0.0f;
        kb = __any((bits_b & 0x7fffffffu) != 0) ? 1.0f : 0.0f;
    };
    auto project_pair = [&](v2f* z, const float (&fa)[9], const float (&fb)[9], bool two, float ka, float kb) {
        if (two) {
            middle_two_frames(z, fa, fb, ka, kb);                     // 1/M of the inverse transform: synthesis window
        } else {
            float mb[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) mb[u] = 0.0f;
            middle_two_frames(z, fa, mb, ka, kb);
        }
    };

    // The first H samples of a frame are complete after it and leave for global memory (outv[j] = position l + 64 j).
    // The stores go through a buffer descriptor whose range is exactly [sf, sf + H): the hardware's bounds check drops
    // the lanes past it, no branch and no per-lane condition.  On a seam (the first N - H samples of the run's region,
    // three frames of a run at hop 275) the samples below the seam's end are ADDED atomically instead, chosen per lane.
    constexpr int kFin = HT ? (HT + 63) / 64 : 16;                     // registers that can hold complete samples
    const long long seam_end = s_lo + ovl;
    auto emit = [&](long long sf, const float (&outv)[kFin]) {
        typedef unsigned u32;
        if (seamed && sf < seam_end) {                                 // wave-uniform
            const int d = (int)(seam_end - sf < H ? seam_end - sf : H);       // positions below d lie on the seam
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xo + sf), 0, 4 * H, 0x00020000);
#pragma unroll
            for (int j = 0; j < kFin; ++j) {
                if (l + 64 * j < d) (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(outv[j], rs, 4 * l + 256 * j, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(u32, outv[j]), rs, 4 * l + 256 * j, 0, 0);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xo + sf), 0, 4 * H, 0x00020000);
#pragma unroll
            for (int j = 0; j < kFin; ++j)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(u32, outv[j]), rs, 4 * l + 256 * j, 0, 0);
        }
    };

    // ---- overlap-add of one frame (component C of z: 0 = frame a, 1 = frame b) starting at sample sf ----
    auto ola = [&](const v2f* z, auto comp, long long sf) {
        constexpr int C = decltype(comp)::value;
        // LDS byte address of ring slot (sf + l + 64 j) mod 2048: one add, one and-or (the ring is 8 KB aligned)
        const unsigned t0 = (unsigned)(((int)(sf & (kP1Ring - 1)) + l) * 4);
        auto slot = [&](int j) -> lds_float* {
            unsigned a;      // ((t0 + 256 j) & 8191) | ring: the compiler leaves it as v_and + v_or
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(t0 + 256u * j), "s"((unsigned)(kP2RingBytes - 1)), "v"(ring_v));
            return (lds_float*)(uintptr_t)a;
        };
        if constexpr (HT != 0) {
            float v[16], outv[kFin];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v4f w4;
                if ((j & 3) == 0) w4 = swindow4(j >> 2);
                const int i = l + 64 * j;
                const float zw = (C ? z[j].y : z[j].x) * w4[j & 3];
                // classes of this register, decided at compile time except on the (at most two) straddling ones
                const bool may_fin = 64 * j < HT, all_fin = 64 * j + 63 < HT;
                const bool may_first = 64 * j + 63 >= kN1 - HT, all_first = 64 * j >= kN1 - HT;
                const bool fin = all_fin || (may_fin && i < HT);
                const bool first = all_first || (may_first && i >= kN1 - HT);
                lds_float* q = slot(j);
                if (all_first) v[j] = zw;
                else if (!may_first) v[j] = *q + zw;
                else v[j] = first ? zw : *q + zw;      // the straddling register reads its slot in every lane
                if (may_fin) outv[j < kFin ? j : 0] = v[j];
                if (!all_fin) {
                    if (may_fin) { if (!fin) *q = v[j]; }
                    else *q = v[j];
                }
            }
            emit(sf, outv);
        } else {
            // any hop: every slot read, added and written back -- zero where the sample is complete
            float outv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v4f w4;
                if ((j & 3) == 0) w4 = swindow4(j >> 2);
                lds_float* q = slot(j);
                outv[j] = *q + (C ? z[j].y : z[j].x) * w4[j & 3];
                *q = (l + 64 * j < H) ? 0.0f : outv[j];
            }
            emit(sf, outv);
        }
    };
    using CompA = std::integral_constant<int, 0>;
    using CompB = std::integral_constant<int, 1>;

    // ---- prologue: spectrum of the run's first pair (zB), windowed input of its second (zA) ----
    v2f zA[16], zB[16], xn[16];
    float fa[9], fb[9];
    float kaA = 0.0f, kbA = 0.0f, kaB, kbB;
    Stamps st;
    fetch_x(f0, xn);
    fetch_mag(f0, fa, fb);
    {
        const v2f* tg = p.tab + 2 * l;
        const int g = wave & 3;
        if (wave < 4) {
            const v4f w = {tg[TE(kT2Win + 4 * g)].x, tg[TE(kT2Win + 4 * g + 1)].x, tg[TE(kT2Win + 4 * g + 2)].x,
                           tg[TE(kT2Win + 4 * g + 3)].x};
            wlds[g * 64 + l] = w;
            wslds[g * 64 + l] = w * (1.0f / kM);
        } else {
            const v2f ta = tg[TE(kT2Tw2 + 2 * g)];
            const v2f tb2 = tg[TE(kT2Tw2 + (g < 3 ? 2 * g + 1 : 6))];
            t2lds[g * 64 + l] = (v4f){ta.x, ta.y, tb2.x, tb2.y};
        }
    }
    __syncthreads();
    if (!active) return;                         // wave-uniform; no barrier after this point
    window_pair(f0 + 1 < f1, xn, zB, kaB, kbB);
    const int n2 = f0 + 2 < f1 ? f0 + 2 : f0;                         // clamped: unused when the run has one pair
    fetch_x(n2, xn);
    fft_forward_1024(zB, ex, l, kloB, tb, st);
    window_pair(n2 + 1 < f1, xn, zA, kaA, kbA);

    // (Unrolling by two so that zA and zB swap roles instead of being copied was measured: 68 bytes of spills, same
    // time.)
    int n = f0;
    for (; n + 2 < f1; n += 2) {                                      // pairs n (B, always two frames) and n + 2 (A)
        const long long sn = (long long)n * H;
        const int n4 = n + 4 < f1 ? n + 4 : n;                        // clamped: unused after the run's last pair
        fetch_x(n4, xn);                             // a whole step ahead: x comes from the other XCDs' L2 / MALL
        dft16<false>(zA);                            // A
        twiddle1<false>(zA, tb);
        ex1_put_cols(zA, ex, lp);
        ex1_get_rows(zA, ex, lp);
        project_pair(zB, fa, fb, true, kaB, kbB);    // B
        fetch_mag(n + 2, fa, fb);
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
        dft8<false>(zA);                             // A: spectrum of pair n + 2 complete
        dft8<false>(zA + 8);
        twiddle1<true>(zB, tb);                      // B
        dft16<true>(zB);
        // window, overlap-add: frame a then frame b (frame order, like the reference)
        __builtin_amdgcn_wave_barrier();
        ola(zB, CompA{}, sn);
        __builtin_amdgcn_wave_barrier();
        ola(zB, CompB{}, sn + H);
        __builtin_amdgcn_wave_barrier();
        kaB = kaA;
        kbB = kbA;
#pragma unroll
        for (int j = 0; j < 16; ++j) zB[j] = zA[j];
        window_pair(n4 + 1 < f1, xn, zA, kaA, kbA);
    }
    {   // epilogue: the run's last pair (possibly a lone frame) has no successor to overlap with
        const long long sn = (long long)n * H;
        const bool two = (n + 1 < f1);
        project_pair(zB, fa, fb, two, kaB, kbB);
        fft_inverse_1024(zB, ex, l, kloB, tb, st);
        __builtin_amdgcn_wave_barrier();
        ola(zB, CompA{}, sn);
        __builtin_amdgcn_wave_barrier();
        if (two) ola(zB, CompB{}, sn + H);
        __builtin_amdgcn_wave_barrier();
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
// (reach = floor((N-1)/H) <= 8, i.e. H >= 114: gl_fast_small_form_limit)
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
    const int kloB = l ? 128 - l : 64;

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
    // an all-zero input frame must give an exactly zero spectrum (see k_gl1024_pair2)
    const float ka = __any(a_nz) ? 1.0f : 0.0f;
    const float kb = __any(b_nz) ? 1.0f : 0.0f;

    Stamps st;
    fft_forward_1024(z, ex, l, kloB, tb, st);
    middle_two_frames(z, ma, mb, ka, kb);
    fft_inverse_1024(z, ex, l, kloB, tb, st);

    float* ya = p.y_out + ((long long)b * T + n) * kN1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const v2f w = tb.get(kT2Win + j);
        ya[l + 64 * j] = z[j].x * w.x;
        if (two) ya[kN1 + l + 64 * j] = z[j].y * w.y;
    }
}

// The Griffin-Lim loop at n_fft 1024.  Two forms: the per-pair frames form for small problems (B*T up to
// gl_fast_small_form_limit = 20 frames per CU, unpinned, reach <= 8) and the run kernel for everything else.
// Measured at T = 641, hop 275 (us per iteration, per-pair form vs run kernel): B=1 13.5 / 20.0, B=4 19.3 / 23.8,
// B=6 20.6 / 25.2, B=8 24.9 / 25.4, B=12 33.4 / 27.8, B=16 42.7 / 33.6: level at about 5 000 frames.
}  // namespace

int gl_fast1024_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                    hipStream_t st) {
    const bool post = gl_post_needed(p);
    int rc = 0;
    // Too few frames to fill the GPU with runs: one wave per frame pair, state kept as frames (k_gl1024_frames).
    // Not with a pinned run length (callers who pin it compare bits with the run kernel) nor with the optional
    // per-iteration signal transforms.
    if (!post && p->run_frames == 0 && (long long)B * T <= gl_fast_small_form_limit(p)) {
        const int reach = (kN1 - 1) / p->hop;
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
        return gl_fast_ola_frames(ybuf[(iters - 1) & 1], x_out, B, T, kN1, p->hop, st);
    }
    const int G = gl_fast_pick_run_length(p, B, T, kP2Waves);
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
    // seams by atomics onto cleared regions; three signal buffers rotate (read, write, clear)
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
        const dim3 grid((unsigned)(((long long)nruns * B + kP2Waves - 1) / kP2Waves)), block(64 * kP2Waves);
#if defined(GL_EXPERIMENT) && defined(GL_P2_GENERIC)
        if (false) {}
#else
        if (p->hop == 275) hipLaunchKernelGGL(k_gl1024_pair2<275>, grid, block, 0, st, aa);   // params.py's hop
#endif
        else hipLaunchKernelGGL(k_gl1024_pair2<0>, grid, block, 0, st, aa);
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
