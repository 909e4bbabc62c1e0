// Kernel-study instrumentation, compiled in only with -DGL_EXPERIMENT -DGL_STAMPS (abl/ builds); the product library
// gets the empty Stamps below and contains none of it.
//
// Phase timing: every phase boundary drains the LDS queue and reads the shader clock; lane 0 of every wave
// adds its per-phase totals to g_stamp at exit (read back with hipMemcpyFromSymbol by the experiment's
// driver).  The drain serialises what the normal build overlaps, so the sum exceeds the normal run time.
#pragma once
#if defined(GL_EXPERIMENT) && defined(GL_STAMPS)
__device__ unsigned long long g_stamp[24];
struct Stamps {
    unsigned long long prev;
    unsigned acc[20];
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int i = 0; i < 20; ++i) acc[i] = 0;
        prev = now();
    }
    __device__ __forceinline__ static unsigned long long now() {
        unsigned long long t;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
        return t;
    }
    __device__ __forceinline__ void mark(int i) {
        unsigned long long t = now();
        acc[i] += (unsigned)(t - prev);
        prev = t;
    }
    __device__ __forceinline__ void flush(int lane) {
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 20; ++i) atomicAdd(&g_stamp[i], (unsigned long long)acc[i]);
            atomicAdd(&g_stamp[20], 1ull);
        }
    }
};
#else
struct Stamps {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int) {}
};
#endif
