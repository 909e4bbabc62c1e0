// Microbenchmark: VALU issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32 at 1/2/4 waves per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N_ITERS 2000
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
    v2f r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (v2f){a + i, b + threadIdx.x};
    v2f va = {a, a}, vb = {b, b};
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i].x) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i].y) : "v"(a), "v"(b)); }
            if (KIND == 1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(va), "v"(vb)); asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(va), "v"(vb)); }
            if (KIND == 2) { asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(r[i]) : "v"(va)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(r[i]) : "v"(va)); }
            if (KIND == 3) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[i].x) : "v"(a)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[i].y) : "v"(a)); }
            if (KIND == 4) { asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(va)); asm volatile("v_mov_b32 %0, %1" : "+v"(r[i].x) : "v"(a)); }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, int waves_per_simd) {
    // 256 CUs, block = 256 threads = 4 waves (one per SIMD); blocks per CU = waves_per_simd
    int blocks = 256 * waves_per_simd;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_wave = (double)N_ITERS * 16;
    // cycles per instruction per SIMD at 2.4 GHz nominal: time * f / (instr_per_wave * waves_per_simd)
    double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);
    printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f cycles/instr/SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_pk_add_f32", w);
        run<3>("v_add_f32", w);
        run<4>("pk_mul+mov", w);
    }
    return 0;
}
