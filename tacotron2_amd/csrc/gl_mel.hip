// Mel -> linear-frequency inversion on the matrix cores (gfx950 fp32-input MFMA).
//
// Replaces griffin_lim.py:18-19:   np.dot(filterbank.T, mel).T
//     mag[b][t][f] = sum_m mel[b][m][t] * filterbank[m][f]
// i.e. C[T x F] = A[T x K] * Bm[K x F] with A[t][m] = mel[b][m][t] (contiguous in t)
// and Bm = the filterbank as stored (contiguous in f); K = n_mels (80).
//
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; the accumulator holds col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  With i = t and j = f the C stores are contiguous
// across lanes and the output is written directly in the (B, T, F) row-major layout the
// Griffin-Lim kernels read.  The result is bit-for-bit an fp32 fma chain over m = 0..K-1.
// The filterbank tile is staged in LDS once per workgroup, operand A goes to registers
// per frame group (see k_mel_to_mag).  The forward product of the analysis path
// (k_mag_to_logmel, N3) is in this file too.
#include "gl_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTileT = 32;      // rows (t) per wave
constexpr int kNJ = 3;          // accumulators (32 bins each) per wave: F = 1025 is 33 blocks of 32 = 11 tiles of 3
constexpr int kTileF = 32 * kNJ;
// PACKED output (gl_mel_run): the register-ring Griffin-Lim kernels read a frame's magnitudes as [quad q][lane][4] =
// (m[k], m[1024-k]) of the lane's mirror pairs u = 2q, 2q+1, then 64 copies of m[512], pre-scaled by 1/2048
// (gl_fast2048.hip, k_mag_repack).  A tile is then 32 lanes x one quad = FOUR accumulators whose registers are the
// four floats of one 16-byte store; 2 lane halves x 4 quads = 8 tiles of 128 columns.  Bin 512 is ONE column: the
// eight tiles of a frame group take turns at it (a fifth accumulator pass with the column broadcast over the 32
// MFMA columns, 33 column blocks of work for the 33 that exist; a ninth tile of its own did 36).
constexpr int kNJPacked = 4;
constexpr int kPackedTiles = 8;
constexpr int kPackRow = 1088;  // floats per packed frame (gl_fft1024.h: kMagPack)
constexpr float kPackScale = 1.0f / 2048.0f;
// bin of column c (0..3) of lane-column jj (0..31) of packed tile ft
__device__ __forceinline__ int packed_bin(int ft, int jj, int c) {
    const int l = 32 * (ft & 1) + jj, u = 2 * (ft >> 1) + (c >> 1);
    const int k = l ? l + 128 * u : (u < 4 ? 128 * u : 64 + 128 * (u - 4));   // lane 0 owns the multiples of 64
    return (c & 1) ? 1024 - k : k;
}
constexpr int kWaves = 4;       // waves per workgroup, stacked along t
constexpr int kKC = 80;         // mel bands per LDS stage of the filterbank tile (40 KB)
// Resident workgroups per CU (3 waves per SIMD).  Measured on config 2 (64 x 861 frames, 80 bands): 109 us; 4 per
// CU 105 us, 2 per CU with the next group's operand A prefetched into registers 123 us, 3 with prefetch 109 us
// -- the other waves of the SIMD already cover the load latency.  Stores alone 44 us, MFMAs alone 84 us.
constexpr int kWgPerCu = 3;

// Buffer addressing: a wave-uniform base in a resource descriptor (scalar registers), one 32-bit lane
// offset shared by every access of the tile, and a scalar offset per access -- the 40 operand loads and
// 48 stores of a tile then need two address VGPRs instead of a 64-bit pointer pair each.
constexpr int kRsrcBytes = 0x7fffffff;   // every access is in range by construction; no bounds masking
constexpr int kRsrcFlags = 0x00020000;   // gfx9 raw buffer, 32-bit data format
__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ void st_buf(float v, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}

// LOGMEL = false: mel is [B][K][T] linear amplitude (what save_mel_to_wav receives).
// LOGMEL = true : mel is [B][T][K] natural-log amplitude, the decoder's output layout; the kernel
//                 applies dynamic_range_decompression (data.py:84-85: exp(mel) / compression_factor)
//                 and the transpose of infer.py:27-28 on the fly, so no (80, T) array is materialised.
//
// A workgroup owns ONE tile of 96 bins for its whole life and walks over (utterance, 128-frame block)
// groups: the filterbank tile (kKC bands x 96 bins) is staged in LDS once, laid out [k][j][4] so the
// three B values a lane feeds to its three MFMAs are one ds_read_b128, and operand A of the NEXT group
// (kKC/2 independent global loads) is fetched into registers while the MFMAs of the current one run, so
// the MFMA loop touches no global memory.  n_mels > kKC falls back to restaging per group and chunk.
//
// XCD-aware order.  Workgroups go to the 8 XCDs round-robin by linear id, and a 128-byte store into the
// 4100-byte rows rarely covers a whole cache line: xcd = id % 8 selects the groups (g % 8 == xcd), so
// all F-tiles of one group run on one XCD, at the same pace, and their partial lines merge in that
// XCD's L2 instead of leaving as partial writes from several.
// FULLK: n_mels == kKC (80, the reference's n_mel_channels): one stage of B for the workgroup's life and a
// branch-free K loop.  Otherwise B is restaged per group and chunk and every band is range-checked.
// PACKED: F = 1025 only; `mag` is the packed [B][T][1088] array and `frames` (null, or [B]) zeroes the rows of
// frames t >= frames[b] as gl_mask_frames does on the plain layout.
template <bool LOGMEL, bool FULLK, bool PACKED>
__global__ void __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(kWgPerCu, kWgPerCu)))
k_mel_to_mag(const float* __restrict__ fb, const float* __restrict__ mel, float* __restrict__ mag, int K, int F,
             int T, float inv_cf, float floor_val, int n_ftiles, int n_tblocks, int n_groups, int n_slots,
             const int* __restrict__ frames) {
    constexpr int NJ = PACKED ? kNJPacked : kNJ;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // tells the compiler it is uniform
    const int li = lane & 31;
    const int lk = lane >> 5;
    const int id = blockIdx.x;
    const int q = id >> 3;
    const int ft = q % n_ftiles;
    const int slot = q / n_ftiles;
    const int f0 = ft * kTileF;
    const int g_first = slot * 8 + (id & 7);
    const int g_step = n_slots * 8;

    __shared__ __attribute__((aligned(16))) v4f Bs[kKC][32];
    __shared__ float Bs512[kKC];                       // PACKED: column 512 of the filterbank
    auto stage_b = [&](int kc) {
        if (PACKED && threadIdx.x < kKC) Bs512[threadIdx.x] = kc + (int)threadIdx.x < K ? fb[(long long)(kc + threadIdx.x) * F + 512] : 0.0f;
        for (int e = threadIdx.x; e < kKC * 32; e += 64 * kWaves) {
            const int k = e >> 5, jj = e & 31;
            const bool k_ok = kc + k < K;
            const float* row = fb + (long long)(k_ok ? kc + k : 0) * F;
            v4f v;
            v[3] = 0.0f;
#pragma unroll
            for (int c = 0; c < NJ; ++c) {
                const int f = PACKED ? packed_bin(ft, jj, c) : f0 + 32 * c + jj;
                v[c] = (k_ok && f < F) ? row[f] : 0.0f;
            }
            Bs[k][jj] = v;
        }
    };
    // Operand A of group g, bands kc .. kc+kKC-1: lane (li, lk) holds band kc + 2 s + lk of frame t0 + li.
    // Addresses are a wave-uniform base (scalar registers) plus one 32-bit lane offset shared by all loads.
    auto load_a = [&](int g, int kc, float (&av)[kKC / 2]) {
        const int b = g / n_tblocks;
        const int t0 = ((g - b * n_tblocks) * kWaves + wave) * kTileT;
        const bool t_ok = t0 + li < T;
        const int lt = t_ok ? li : 0;                        // lanes past T read frame t0 and are zeroed
        if (g >= n_groups || t0 >= T) {                      // wave-uniform: nothing of this tile exists, read nothing
#pragma unroll
            for (int s = 0; s < kKC / 2; ++s) av[s] = 0.0f;
            return;
        }
        const float* base = LOGMEL ? mel + ((long long)b * T + t0) * K + kc : mel + ((long long)b * K + kc) * T + t0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, kRsrcBytes, kRsrcFlags);
        const int k_stride = LOGMEL ? 1 : T;
        const int lane_off = (LOGMEL ? lt * K + lk : lk * T + lt) * 4;
#pragma unroll
        for (int s = 0; s < kKC / 2; ++s) {
            if (FULLK) {
                const float a = LOGMEL ? ld_buf(rs, lane_off + 8 * s, 0) : ld_buf(rs, lane_off, 8 * s * k_stride);
                av[s] = t_ok ? (LOGMEL ? expf(a) * inv_cf : a) : 0.0f;
            } else {
                // kv = bands left from 2 s on (wave-uniform): >= 2 both half-waves load, 1 only lk = 0 has a
                // band (the others read that band too and are zeroed), <= 0 nothing is read.
                const int kv = K - kc - 2 * s;
                float a = 0.0f;
                if (kv >= 2) a = LOGMEL ? ld_buf(rs, lane_off + 8 * s, 0) : ld_buf(rs, lane_off, 8 * s * k_stride);
                else if (kv == 1) a = LOGMEL ? ld_buf(rs, lane_off - 4 * lk + 8 * s, 0)
                                             : ld_buf(rs, lane_off - 4 * lk * k_stride, 8 * s * k_stride);
                av[s] = (t_ok && lk < kv) ? (LOGMEL ? expf(a) * inv_cf : a) : 0.0f;
            }
        }
    };
    auto mfma_chunk = [&](f32x16 (&acc)[NJ], const float (&av)[kKC / 2], int kend) {
#pragma unroll
        for (int s = 0; s < kKC / 2; ++s) {
            if (FULLK || 2 * s < kend) {                     // wave-uniform; rows past K are zero in Bs and av
                const v4f bv = Bs[2 * s + lk][li];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[j], acc[j], 0, 0, 0);
            }
            // Without a fence the scheduler hoists all 40 ds_read_b128 (160 VGPRs) above the first MFMA.
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto store = [&](int g, const f32x16 (&acc)[NJ]) {
        const int b = g / n_tblocks;
        const int t0 = ((g - b * n_tblocks) * kWaves + wave) * kTileT;
        if (t0 >= T) return;
        if constexpr (PACKED) {
            // registers r of the four accumulators = the quad (m[k], m[1024-k], m[k'], m[1024-k']) of lane
            // 32 (ft & 1) + li for frame t0 + row(r): one 16-byte store, 512 contiguous bytes per half-wave
            float* base = mag + ((long long)b * T + t0) * kPackRow + (ft >> 1) * 256 + (ft & 1) * 128;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, kRsrcBytes, kRsrcFlags);
            const int lane_off = (4 * lk * kPackRow + 4 * li) * 4;
            const int t_end = frames ? min(frames[b], T) : T;         // rows from t_end on are written as zeros
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const int t = t0 + ro + 4 * lk;
                if (t >= T) continue;
                const bool live = t < t_end;     // a select, not a product: padding rows are exact zeros whatever the
                typedef unsigned v4u __attribute__((ext_vector_type(4)));   // decoder left there (inf * 0 would be NaN)
                v4f v;
#pragma unroll
                for (int j = 0; j < NJ; ++j) v[j] = live ? fmaxf(acc[j][r], floor_val) * kPackScale : 0.0f;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs, lane_off, ro * kPackRow * 4, 0);
            }
            return;
        }
        float* base = mag + ((long long)b * T + t0) * F + f0;   // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, kRsrcBytes, kRsrcFlags);
        const int lane_off = (4 * lk * F + li) * 4;
        const bool full = t0 + kTileT <= T && f0 + kTileF <= F;
        if (full) {
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st_buf(fmaxf(acc[j][r], floor_val), rs, lane_off, (((r & 3) + 8 * (r >> 2)) * F + 32 * j) * 4);
        } else {
#pragma unroll
            for (int j = 0; j < kNJ; ++j) {
                if (f0 + 32 * j + li >= F) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (t0 + ro + 4 * lk < T)
                        st_buf(fmaxf(acc[j][r], floor_val), rs, lane_off, (ro * F + 32 * j) * 4);  // -inf: no clamp
                }
            }
        }
    };
    auto zero = [&](f32x16 (&acc)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    };

    // PACKED: bin 512 of the group's frames, 64 copies each at [1024, 1088) of the packed row.  All 32 MFMA columns
    // carry the same filterbank column, so every lane of a half-wave holds m[512] of its register's frame and
    // lanes li < 16 store four copies each (256 contiguous bytes per frame).
    auto mfma_512 = [&](f32x16& a5, const float (&av)[kKC / 2], int kend) {
#pragma unroll
        for (int s = 0; s < kKC / 2; ++s)
            if (FULLK || 2 * s < kend) a5 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], Bs512[2 * s + lk], a5, 0, 0, 0);
    };
    auto store_512 = [&](int g, const f32x16& a5) {
        const int b = g / n_tblocks;
        const int t0 = ((g - b * n_tblocks) * kWaves + wave) * kTileT;
        if (t0 >= T || li >= 16) return;
        float* base = mag + ((long long)b * T + t0) * kPackRow + 1024;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, kRsrcBytes, kRsrcFlags);
        const int lane_off = (4 * lk * kPackRow + 4 * li) * 4;
        const int t_end = frames ? min(frames[b], T) : T;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            const int t = t0 + ro + 4 * lk;
            if (t >= T) continue;
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const float m = t < t_end ? fmaxf(a5[r], floor_val) * kPackScale : 0.0f;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, (v4f){m, m, m, m}), rs, lane_off,
                                                   ro * kPackRow * 4, 0);
        }
    };
    // the tiles of one (slot, XCD) walk the same groups: the one whose turn it is on this step takes bin 512
    int turn = (slot + (id & 7)) % kPackedTiles;

    if (FULLK) {
        stage_b(0);
        __syncthreads();
        for (int g = g_first; g < n_groups; g += g_step) {
            float av[kKC / 2];
            load_a(g, 0, av);
            f32x16 acc[NJ];
            zero(acc);
            mfma_chunk(acc, av, K);
            store(g, acc);
            if constexpr (PACKED) {
                if (turn == ft) {                    // workgroup-uniform
                    f32x16 a5;
#pragma unroll
                    for (int r = 0; r < 16; ++r) a5[r] = 0.0f;
                    mfma_512(a5, av, K);
                    store_512(g, a5);
                }
                turn = turn + 1 == kPackedTiles ? 0 : turn + 1;
            }
        }
    } else {
        for (int g = g_first; g < n_groups; g += g_step) {   // workgroup-uniform trip count
            f32x16 acc[NJ];
            zero(acc);
            f32x16 a5;
#pragma unroll
            for (int r = 0; r < 16; ++r) a5[r] = 0.0f;
            for (int kc = 0; kc < K; kc += kKC) {
                float av[kKC / 2];
                load_a(g, kc, av);
                __syncthreads();
                stage_b(kc);
                __syncthreads();
                mfma_chunk(acc, av, min(kKC, K - kc));
                if (PACKED && turn == ft) mfma_512(a5, av, min(kKC, K - kc));
            }
            store(g, acc);
            if constexpr (PACKED) {
                if (turn == ft) store_512(g, a5);
                turn = turn + 1 == kPackedTiles ? 0 : turn + 1;
            }
        }
    }
}

// N3  data.py:112-114: mel = filterbank @ |stft|, log(clip(mel, 0.01) * compression_factor).
//     logmel[b][m][t] = log(max(sum_f fb[m][f] * mag[b][t][f], clip_min) * cf)
// C[m x t] = A[m x f] * B[f x t] with i = m, j = t: the accumulator's 32 lanes of one register are 32
// consecutive t of one mel band, so the (B, n_mels, T) output is stored coalesced.  One wave owns 32
// frames and all mel bands (TILES accumulators of 32 bands); K = F runs two bins per MFMA.
//
// Both operands are contiguous in f, the contraction index, so they are staged through LDS: per chunk of
// kFC bins the workgroup copies fb[0..32 TILES)[f0..f0+kFC) and mag[t0..t0+128)[f0..f0+kFC) with
// 16-byte loads (8 lanes cover the 128 bytes of one row's chunk) into rows padded to kFC + 1 floats, which
// makes the MFMA operand reads (32 rows at one f) conflict-free.  The global loads of chunk c + 1 are
// issued before the MFMAs of chunk c and written to LDS after them.
constexpr int kFC = 32;                       // bins per chunk
constexpr int kFS = kFC + 1;                  // LDS row stride in floats
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));   // rows are 4 F bytes apart: 4-byte aligned
template <int TILES>
__global__ void __launch_bounds__(64 * kWaves) k_mag_to_logmel(const float* __restrict__ fb,
                                                               const float* __restrict__ mag,
                                                               float* __restrict__ logmel, int n_mels, int F, int T,
                                                               float clip_min, float cf) {
    constexpr int kRowsA = 32 * TILES;        // filterbank rows staged (rows >= n_mels are zero)
    constexpr int kRowsB = 32 * kWaves;       // frames staged
    constexpr int kQuads = (kRowsA + kRowsB) * (kFC / 4) / (64 * kWaves);   // 16-byte pieces per thread and chunk
    static_assert((kRowsA + kRowsB) * (kFC / 4) % (64 * kWaves) == 0, "staging must divide evenly");
    __shared__ float As[kRowsA * kFS];
    __shared__ float Bs[kRowsB * kFS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int t_blk = blockIdx.x * kRowsB;    // first frame of the workgroup
    const int li = lane & 31;
    const int lk = lane >> 5;
    const float* mag_b = mag + (long long)b * T * F;

    // Piece q of this thread: row (A rows first, then B rows) and the quad of 4 bins inside the chunk.
    auto piece = [&](int q, int& row, int& c4) {
        const int e = q * 64 * kWaves + threadIdx.x;
        row = e >> 3;
        c4 = (e & 7) * 4;
    };
    auto fetch = [&](int f0, v4f (&regs)[kQuads]) {
#pragma unroll
        for (int q = 0; q < kQuads; ++q) {
            int row, c4;
            piece(q, row, c4);
            const bool is_a = row < kRowsA;                  // uniform per wave: kRowsA is a multiple of 8 rows x 8 lanes
            const int r = is_a ? row : row - kRowsA;
            const bool row_ok = is_a ? r < n_mels : t_blk + r < T;
            const float* src = is_a ? fb + (long long)r * F : mag_b + (long long)(t_blk + r) * F;
            const int f = f0 + c4;
            v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (row_ok) {
                if (f + 4 <= F) {
                    v = *reinterpret_cast<const v4f_u*>(src + f);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (f + c < F) v[c] = src[f + c];
                }
            }
            regs[q] = v;
        }
    };
    auto deposit = [&](const v4f (&regs)[kQuads]) {
#pragma unroll
        for (int q = 0; q < kQuads; ++q) {
            int row, c4;
            piece(q, row, c4);
            float* dst = row < kRowsA ? As + row * kFS + c4 : Bs + (row - kRowsA) * kFS + c4;
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[c] = regs[q][c];
        }
    };

    f32x16 acc[TILES];
#pragma unroll
    for (int j = 0; j < TILES; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    v4f regs[kQuads];
    fetch(0, regs);
    for (int f0 = 0; f0 < F; f0 += kFC) {
        __syncthreads();                                     // the previous chunk has been consumed
        deposit(regs);
        __syncthreads();
        if (f0 + kFC < F) fetch(f0 + kFC, regs);             // in flight during the MFMAs below
        const float* bp = Bs + (wave * 32 + li) * kFS + lk;
        const float* ap = As + li * kFS + lk;
#pragma unroll
        for (int k = 0; k < kFC; k += 2) {
            const float bv = bp[k];
#pragma unroll
            for (int j = 0; j < TILES; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[j * 32 * kFS + k], bv, acc[j], 0, 0, 0);
        }
    }

    const int t = t_blk + wave * 32 + li;
    if (t >= T) return;
    float* out = logmel + (long long)b * n_mels * T;
#pragma unroll
    for (int j = 0; j < TILES; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < n_mels) out[(long long)m * T + t] = logf(fmaxf(acc[j][r], clip_min) * cf);
        }
    }
}

}  // namespace

int gl_mag_to_logmel_launch(const float* fb, const float* mag, float* logmel, int n_mels, int F, int B, int T,
                            float clip_min, float compression_factor, hipStream_t st) {
    dim3 grid((T + 32 * kWaves - 1) / (32 * kWaves), B);
    const int tiles = (n_mels + 31) / 32;
    if (tiles < 1 || tiles > 4) return gl_set_error(GL_ERR_INVALID, "mag_to_logmel: n_mels=%d outside [1, 128]", n_mels);
    auto kern = tiles == 1 ? k_mag_to_logmel<1> : tiles == 2 ? k_mag_to_logmel<2> : tiles == 3 ? k_mag_to_logmel<3>
                                                                                               : k_mag_to_logmel<4>;
    hipLaunchKernelGGL(kern, grid, dim3(64 * kWaves), 0, st, fb, mag, logmel, n_mels, F, T, clip_min,
                       compression_factor);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_mel_launch(const float* fb, const float* mel, float* mag, int n_mels, int F, int B, int T, bool logmel,
                  float compression_factor, bool clamp, int n_cu, hipStream_t st, bool packed, const int* frames) {
    const float floor_val = clamp ? 0.0f : -INFINITY;     // GL_OPT_CLAMP_MAG (pseudo-inverse bases go negative)
    if (packed && F != 1025) return gl_set_error(GL_ERR_INVALID, "packed magnitudes exist for n_fft 2048 only");
    const int n_ftiles = packed ? kPackedTiles : (F + kTileF - 1) / kTileF;
    const int n_tblocks = (T + kTileT * kWaves - 1) / (kTileT * kWaves);
    const int n_groups = n_tblocks * B;
    // One round of resident workgroups: per XCD (n_cu / 8 CUs) kWgPerCu per CU, shared by the F-tiles.
    int n_slots = (n_cu / 8) * kWgPerCu / n_ftiles;
    if (n_slots < 1) n_slots = 1;
    if (n_slots > (n_groups + 7) / 8) n_slots = (n_groups + 7) / 8;
    dim3 grid((unsigned)(8 * n_ftiles * n_slots));
    const float inv_cf = logmel ? 1.0f / compression_factor : 1.0f;
    const bool fullk = n_mels == kKC;
    auto kern = packed ? (logmel ? (fullk ? k_mel_to_mag<true, true, true> : k_mel_to_mag<true, false, true>)
                                 : (fullk ? k_mel_to_mag<false, true, true> : k_mel_to_mag<false, false, true>))
                       : (logmel ? (fullk ? k_mel_to_mag<true, true, false> : k_mel_to_mag<true, false, false>)
                                 : (fullk ? k_mel_to_mag<false, true, false> : k_mel_to_mag<false, false, false>));
    hipLaunchKernelGGL(kern, grid, dim3(64 * kWaves), 0, st, fb, mel, mag, n_mels, F, T, inv_cf, floor_val, n_ftiles,
                       n_tblocks, n_groups, n_slots, frames);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}
