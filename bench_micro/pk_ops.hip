// Checks the packed-fp32 modifier forms (op_sel / neg_lo / neg_hi) that gl_fast.hip issues
// through one-instruction inline asm against plain scalar arithmetic.  Build:
//   hipcc -O2 --offload-arch=gfx950 -o pk_ops pk_ops.hip && ./pk_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#include "../tacotron2_amd/csrc/gl_pk.h"

__global__ void k(const v2f* a, const v2f* b, const v2f* c, v2f* o) {
    int i = threadIdx.x;
    v2f A = a[i], B = b[i], C = c[i];
    o[i + 64 * 0] = pk_addi(A, B);
    o[i + 64 * 1] = pk_subi(A, B);
    o[i + 64 * 2] = pk_addc(A, B);
    o[i + 64 * 3] = pk_subc(A, B);
    o[i + 64 * 4] = cmul(A, B);
    o[i + 64 * 5] = cmulc(A, B);
    o[i + 64 * 6] = cmul_add(A, B, C);
    o[i + 64 * 7] = cmul_sub(A, B, C);
    o[i + 64 * 8] = cmulc_add(A, B, C);
    o[i + 64 * 9] = cmulc_subconj(A, B, C);
    o[i + 64 * 10] = pk_swapc_add(A, B);
}
static void ref(int op, const float* A, const float* B, const float* C, float* r) {
    float ax = A[0], ay = A[1], bx = B[0], by = B[1], cx = C[0], cy = C[1];
    switch (op) {
    case 0: r[0] = ax - by; r[1] = ay + bx; break;                 // a + i b
    case 1: r[0] = ax + by; r[1] = ay - bx; break;                 // a - i b
    case 2: r[0] = ax + bx; r[1] = ay - by; break;                 // a + conj b
    case 3: r[0] = ax - bx; r[1] = ay + by; break;                 // a - conj b
    case 4: r[0] = ax * bx - ay * by; r[1] = ax * by + ay * bx; break;
    case 5: r[0] = ax * bx + ay * by; r[1] = ay * bx - ax * by; break;
    case 6: r[0] = cx + ax * bx - ay * by; r[1] = cy + ax * by + ay * bx; break;      // c + a*b
    case 7: r[0] = cx - (ax * bx - ay * by); r[1] = cy - (ax * by + ay * bx); break;  // c - a*b
    case 8: r[0] = cx + ax * bx + ay * by; r[1] = cy + ay * bx - ax * by; break;      // c + a*conj b
    case 9: r[0] = cx - (ax * bx + ay * by); r[1] = -(cy - (ay * bx - ax * by)); break; // conj(c - a*conj b)
    case 10: r[0] = ax + by; r[1] = -ay + bx; break;               // conj(a) + swap(b)
    }
}
int main() {
    const int n = 64, nops = 11;
    float ha[2 * n], hb[2 * n], hc[2 * n], ho[2 * n * nops];
    for (int i = 0; i < 2 * n; ++i) { ha[i] = sinf(1.0f + i) * 3; hb[i] = cosf(0.3f * i + 2) * 2; hc[i] = sinf(0.7f * i) + 0.1f * i; }
    v2f *a, *b, *c, *o;
    hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&c, sizeof(hc)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
    hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(c, hc, sizeof(hc), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, a, b, c, o);
    if (hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
    int bad = 0;
    for (int op = 0; op < nops; ++op) {
        float worst = 0;
        for (int i = 0; i < n; ++i) {
            float r[2];
            ref(op, ha + 2 * i, hb + 2 * i, hc + 2 * i, r);
            for (int h = 0; h < 2; ++h) worst = fmaxf(worst, fabsf(r[h] - ho[(op * 64 + i) * 2 + h]));
        }
        printf("op %d max err %g %s\n", op, worst, worst < 1e-5f ? "ok" : "BAD");
        bad += worst >= 1e-5f;
    }
    return bad ? 1 : 0;
}
