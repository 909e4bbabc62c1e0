// Packed-fp32 complex primitives for gfx950, one v_pk_* instruction each.
//
// A complex number is a v2f (re, im) in an aligned VGPR pair.  hipcc folds whole-vector
// negation and the (x,y)->(y,x) swap into VOP3P modifiers, but not a negation of ONE half, so
// multiplication by +-i, conjugation and the second half of a complex multiply cost it a
// v_xor + v_mov pair each (a quarter of the register-ring kernel's VALU instructions).  The
// forms below state the modifiers outright.  op_sel picks the source half feeding the LOW
// result, op_sel_hi the half feeding the HIGH result; neg_lo / neg_hi negate the operand of the
// low / high result.  Each asm is a single instruction, so the compiler keeps scheduling them
// and keeps inserting the wait state a dependent instruction needs after a packed op
// (bench_micro/pk_ops.hip checks every form against scalar arithmetic on the GPU).
#pragma once

#define GL_PK2(name, text)                                                          \
    __device__ __forceinline__ v2f name(v2f a, v2f b) {                             \
        v2f r;                                                                      \
        asm(text : "=v"(r) : "v"(a), "v"(b));                                       \
        return r;                                                                   \
    }
#define GL_PK3(name, text)                                                          \
    __device__ __forceinline__ v2f name(v2f a, v2f b, v2f c) {                      \
        v2f r;                                                                      \
        asm(text : "=v"(r) : "v"(a), "v"(b), "v"(c));                               \
        return r;                                                                   \
    }

GL_PK2(pk_addi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")   // a + i*b
GL_PK2(pk_subi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")   // a - i*b
GL_PK2(pk_addc, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                                // a + conj(b)
GL_PK2(pk_subc, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                                // a - conj(b)
GL_PK2(pk_subcc, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]")                   // conj(a - b)
GL_PK2(pk_swapc_add, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]")  // conj(a) + (b.y, b.x)

// halves of a complex multiply: (a.x * w) and (a.y * i*w) [+ c]
GL_PK2(pk_mul_xw, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")                                       // a.x * (w.x,  w.y)
GL_PK2(pk_mul_xwc, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]")                         // a.x * (w.x, -w.y)
GL_PK3(pk_fma_xw, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]")                                 // c + a.x*(w.x, w.y)
GL_PK3(pk_fma_nxw, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")  // c - a.x*(w.x, w.y)
GL_PK3(pk_fma_xwc, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]")                 // c + a.x*(w.x,-w.y)
GL_PK3(pk_fma_nxw_cc, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,1]")  // conj(c) - a.x*(w.x, w.y)
GL_PK3(pk_fma_yiw, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]")  // c + a.y*(-w.y, w.x)
GL_PK3(pk_fma_nyiw, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]") // c - a.y*(-w.y, w.x)
GL_PK3(pk_fma_ywcs, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]")                // c + a.y*(w.y, w.x)

__device__ __forceinline__ v2f cmul(v2f a, v2f w) { return pk_fma_yiw(a, w, pk_mul_xw(a, w)); }          // a*w
__device__ __forceinline__ v2f cmulc(v2f a, v2f w) { return pk_fma_ywcs(a, w, pk_mul_xwc(a, w)); }       // a*conj(w)
__device__ __forceinline__ v2f cmul_add(v2f a, v2f w, v2f c) { return pk_fma_yiw(a, w, pk_fma_xw(a, w, c)); }     // c + a*w
__device__ __forceinline__ v2f cmul_sub(v2f a, v2f w, v2f c) { return pk_fma_nyiw(a, w, pk_fma_nxw(a, w, c)); }   // c - a*w
__device__ __forceinline__ v2f cmulc_add(v2f a, v2f w, v2f c) { return pk_fma_ywcs(a, w, pk_fma_xwc(a, w, c)); }  // c + a*conj(w)
// conj(c - a*conj(w)) = conj(c) - conj(a)*w : (c.x - a.x w.x - a.y w.y, -c.y - a.x w.y + a.y w.x)
__device__ __forceinline__ v2f cmulc_subconj(v2f a, v2f w, v2f c) { return pk_fma_yiw(a, w, pk_fma_nxw_cc(a, w, c)); }
