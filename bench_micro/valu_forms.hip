// Microbenchmark: issue cost of the packed-fp32 forms the fused kernels use (gfx950), 2 and 3 waves per SIMD.
// Each kernel runs 8 independent accumulator chains; cycles are per instruction per SIMD at 2.4 GHz nominal.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N_ITERS 4000
#define OP(txt) for (int i = 0; i < 8; ++i) asm volatile(txt : "+v"(r[i]) : "v"(va), "v"(vb))
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
    v2f r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (v2f){a + i, b + threadIdx.x};
    v2f va = {a, b}, vb = {b, a};
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if (KIND == 0) OP("v_pk_add_f32 %0, %0, %1");
            if (KIND == 1) OP("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]");
            if (KIND == 2) OP("v_pk_add_f32 %0, %0, %1 neg_hi:[0,1]");
            if (KIND == 3) OP("v_pk_fma_f32 %0, %1, %2, %0");
            if (KIND == 4) OP("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]");
            if (KIND == 5) OP("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]");
            if (KIND == 6) OP("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]");
            if (KIND == 7) { for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i].x) : "v"(a), "v"(b)); }
            if (KIND == 8) { for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i].x) : "v"(a) : ); }
            if (KIND == 9) { for (int i = 0; i < 8; ++i) asm volatile("v_mov_b64 %0, %1" : "+v"(r[i]) : "v"(va)); }
            if (KIND == 10) { for (int i = 0; i < 8; ++i) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i].x)); }
            if (KIND == 11) { for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i].x) : "v"(a)); }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name) {
    for (int w : {2, 3}) {
        int blocks = 256 * w;
        float* out; (void)hipMalloc(&out, blocks * 256 * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double cyc = ms * 1e-3 * 2.4e9 / ((double)N_ITERS * 16 * w);
        printf("%-44s waves/SIMD %d: %.2f cycles/instr/SIMD\n", name, w, cyc);
        (void)hipFree(out);
    }
}
int main() {
    run<0>("v_pk_add_f32 plain");
    run<1>("v_pk_add_f32 op_sel swap + neg_lo (a + i b)");
    run<2>("v_pk_add_f32 neg_hi (a + conj b)");
    run<3>("v_pk_fma_f32 plain");
    run<4>("v_pk_fma_f32 op_sel_hi:[0,1,1]");
    run<5>("v_pk_fma_f32 op_sel swizzle + neg");
    run<6>("v_pk_mul_f32 op_sel_hi:[0,1]");
    run<7>("v_fma_f32");
    run<8>("v_cndmask_b32");
    run<9>("v_mov_b64");
    run<10>("v_rsq_f32");
    run<11>("v_mul_f32");
    return 0;
}
