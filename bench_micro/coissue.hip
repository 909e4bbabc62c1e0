// Microbenchmark (gfx950): does LDS / VMEM traffic of ONE wave take VALU time away from the OTHER wave of its SIMD?
// 512-thread workgroups, one per CU: waves 0..3 ("A", one per SIMD) run a stream of independent v_pk_fma_f32; waves
// 4..7 ("B", their SIMD partners) run one of: nothing, the same VALU stream, ds_write_b128, ds_write_b64,
// ds_read_b128, global loads (L2-resident).  Reported: shader-clock cycles per v_pk_fma of the A waves (s_memtime
// around their loop, median over workgroups) and the B waves' own loop time.  Also MIXED: every wave interleaves
// VALU and LDS stores itself (the transposes of k_gl2048_regring), against the sum of the two alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define N_ITERS 400
#define UNROLL 16
enum { B_IDLE = 0, B_VALU = 1, B_DSW128 = 2, B_DSW64 = 3, B_DSR128 = 4, B_GLD = 5, MIXED64 = 6, MIXED128 = 7, A_ONLY_LDS = 8 };

template <int KIND>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* ticks, const float* src, float a, float b) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 64 * 4 * 4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    v2f r[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r[i] = (v2f){a + i, b + lane};
    const v2f va = {a, a}, vb = {b, b};
    float* my = lds + wave * 64 * 16 + lane * 4;          // 16-byte slot per lane, 4 slots of 1 KB per wave
    v4f w4 = {a, b, a, b};
    v4f acc4 = {0, 0, 0, 0};
    const bool roleA = wave < 4;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (KIND == MIXED64 || KIND == MIXED128 || KIND == A_ONLY_LDS) {
        // every wave: per iteration UNROLL pk_fma (x4 rounds) and one transpose's worth of stores (16 x b64 or 8 x b128)
        for (int it = 0; it < N_ITERS; ++it) {
            if (KIND != A_ONLY_LDS) {
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int i = 0; i < UNROLL; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(va), "v"(vb));
            }
            if (KIND == MIXED64 || KIND == A_ONLY_LDS) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"((unsigned)(size_t)(my) & 0xffff), "v"(r[i]), "n"(0) : "memory");
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)(my) & 0xffff), "v"(w4) : "memory");
            }
        }
    } else if (roleA || KIND == B_VALU) {
        for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < UNROLL; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(va), "v"(vb));
        }
    } else if (KIND == B_DSW128) {
        for (int it = 0; it < N_ITERS * 4; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)(my) & 0xffff), "v"(w4) : "memory");
        }
    } else if (KIND == B_DSW64) {
        for (int it = 0; it < N_ITERS * 4; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(size_t)(my) & 0xffff), "v"(r[i]) : "memory");
        }
    } else if (KIND == B_DSR128) {
        for (int it = 0; it < N_ITERS * 4; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v4f t;
                asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)(my) & 0xffff) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc4 += t;
            }
        }
    } else if (KIND == B_GLD) {
        const v4f* p = reinterpret_cast<const v4f*>(src) + (blockIdx.x * 512 + threadIdx.x);
        for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc4 += __builtin_nontemporal_load(p + ((it * 8 + i) & 63) * 256 * 512);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc4.x + acc4.y + acc4.z + acc4.w;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += r[i].x + r[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s + lds[threadIdx.x];
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* src) {
    const int blocks = 256;
    float* out; hipMalloc(&out, blocks * 512 * 4);
    unsigned long long* ticks; hipMalloc(&ticks, blocks * 8 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, out, ticks, src, 1.0f, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> A, B;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? A : B).push_back((double)h[b * 8 + w]);
    std::sort(A.begin(), A.end()); std::sort(B.begin(), B.end());
    const double nfma = (double)N_ITERS * 4 * UNROLL;
    printf("%-34s A waves: %9.0f ticks = %.2f per pk_fma | B waves: %9.0f ticks\n", name, A[A.size() / 2], A[A.size() / 2] / nfma, B[B.size() / 2]);
    hipFree(out); hipFree(ticks);
}
int main() {
    float* src; hipMalloc(&src, (size_t)64 * 256 * 512 * 16); hipMemset(src, 0, (size_t)64 * 256 * 512 * 16);
    run<B_IDLE>("B idle", src);
    run<B_VALU>("B same pk_fma stream", src);
    run<B_DSW128>("B ds_write_b128 stream", src);
    run<B_DSW64>("B ds_write_b64 stream", src);
    run<B_DSR128>("B ds_read_b128 stream", src);
    run<B_GLD>("B global_load_dwordx4 nt stream", src);
    run<A_ONLY_LDS>("all waves: 16 ds_write_b64 per it", src);
    run<MIXED64>("all waves: 64 pk_fma + 16 dsw_b64", src);
    run<MIXED128>("all waves: 64 pk_fma + 8 dsw_b128", src);
    return 0;
}
