// Microbenchmark: shader clock actually delivered under packed-fp32 load on every SIMD.
// s_memtime counts shader-clock cycles, s_memrealtime a constant 100 MHz reference; their ratio over a
// long v_pk_fma_f32 loop on 2 waves per SIMD of all CUs is the sustained clock, and instructions issued
// divided by the s_memtime span is the issue cost in real cycles (valu_forms.hip assumes 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N_ITERS 20000
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, float a, float b) {
    v2f r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (v2f){a + i, b + threadIdx.x};
    v2f va = {a, b}, vb = {b, a};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(va), "v"(vb));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i].x) : "v"(a), "v"(b));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(va));
            }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int KIND>
void run(const char* name, int w) {
    int blocks = 256 * w;
    float* out; unsigned long long* clk; unsigned long long h[2];
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&clk, 16);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, 1.0f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / ((double)h[1] / 100e6) * 1e-9;
    double cyc = (double)h[0] / ((double)N_ITERS * 16 * w);
    double kcyc = ms * 1e-3 * ghz * 1e9 / ((double)N_ITERS * 16 * w);
    printf("%-16s waves/SIMD %d: shader clock %.3f GHz; wave 0 span %.2f, whole kernel %.2f shader cycles per instruction per SIMD (kernel %.3f ms, wave 0 %.3f ms)\n",
           name, w, ghz, cyc, kcyc, ms, (double)h[1] / 100e6 * 1e3);
    (void)hipFree(out); (void)hipFree(clk);
}
int main() {
    for (int w : {1, 2, 3, 4}) run<0>("v_pk_fma_f32", w);
    for (int w : {1, 2, 4}) run<1>("v_fma_f32", w);
    for (int w : {2}) run<2>("v_pk_add_f32", w);
    return 0;
}
