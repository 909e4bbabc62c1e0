// Generic Griffin-Lim kernels: any power-of-two N in [64, 4096], any hop.
// One workgroup per frame, N/2-point complex Stockham radix-2 FFT in LDS (the
// real transform is packed as even/odd -> one half-size complex FFT), frames
// materialised in HBM and overlap-added by a gather kernel.  This is the
// reference-shaped, unfused path; gl_fast.hip holds the fused wave-per-frame
// kernels used for the geometries the benchmarks run.
//
// Replaces: audio_utilities.py:91-92 (frame*window, rfft), :181-183 (projection),
//           :112-113 (irfft, window, overlap-add).
#include <math.h>

#include <vector>

#include "gl_internal.h"

namespace {

constexpr int kThreads = 256;

// Stockham autosort radix-2; a holds the input, returns the buffer holding the
// output (natural order).  tw[k] = exp(-2*pi*i*k/N), N = 2*M.
template <bool INV>
__device__ float2* wg_fft(float2* a, float2* b, int M, int N, const float2* __restrict__ tw) {
    const int half = M >> 1;
    for (int p = 1; p < M; p <<= 1) {
        const int tw_step = N / (2 * p);
        for (int i = threadIdx.x; i < half; i += blockDim.x) {
            const int k = i & (p - 1);
            float2 u0 = a[i];
            float2 u1 = a[i + half];
            float2 w = tw[k * tw_step];
            float2 v = INV ? c_mulc(u1, w) : c_mul(u1, w);
            const int j = (i << 1) - k;
            b[j] = c_add(u0, v);
            b[j + p] = c_sub(u0, v);
        }
        __syncthreads();
        float2* t = a;
        a = b;
        b = t;
    }
    return a;
}

__device__ __forceinline__ void load_frame(float2* a, const float* __restrict__ xf, const float* __restrict__ win,
                                           int M) {
    for (int n = threadIdx.x; n < M; n += blockDim.x)
        a[n] = make_float2(win[2 * n] * xf[2 * n], win[2 * n + 1] * xf[2 * n + 1]);
    __syncthreads();
}

// Z (half-size FFT of the packed frame) -> rfft bins X[k], X[M-k] for 0 < k <= M/2
__device__ __forceinline__ void rfft_pair(float2 zk, float2 zm, float2 w, float2& Xk, float2& Xmk) {
    float2 A = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));   // Ze[k]
    float2 Zo = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));  // Zo[k]
    float2 Bv = c_mul(w, Zo);
    Xk = c_add(A, Bv);
    Xmk = c_conj(c_sub(A, Bv));
}

// rfft bins X[k], X[M-k] -> Z[k], Z[M-k] of the packed inverse transform
__device__ __forceinline__ void irfft_pair(float2 Xk, float2 Xmk, float2 w, float2& zk, float2& zm) {
    float2 A = make_float2(0.5f * (Xk.x + Xmk.x), 0.5f * (Xk.y - Xmk.y));
    float2 Bv = make_float2(0.5f * (Xk.x - Xmk.x), 0.5f * (Xk.y + Xmk.y));
    float2 Zo = c_mulc(Bv, w);  // conj(w) * Bv
    zk = make_float2(A.x - Zo.y, A.y + Zo.x);   // A + i*Zo
    zm = make_float2(A.x + Zo.y, -A.y + Zo.x);  // conj(A) + i*conj(Zo)
}

__global__ void __launch_bounds__(kThreads) k_generic_stft(const float* __restrict__ x, long long x_stride,
                                                           float2* __restrict__ S, const float* __restrict__ win,
                                                           const float2* __restrict__ tw, int N, int H, int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* a = reinterpret_cast<float2*>(smem_raw);
    const int M = N >> 1;
    float2* b = a + M;
    const int t = blockIdx.x, bi = blockIdx.y;
    load_frame(a, x + (long long)bi * x_stride + (long long)t * H, win, M);
    float2* Z = wg_fft<false>(a, b, M, N, tw);
    float2* out = S + ((long long)bi * T + t) * (M + 1);
    for (int k = threadIdx.x; k <= M / 2; k += blockDim.x) {
        if (k == 0) {
            float2 z = Z[0];
            out[0] = make_float2(z.x + z.y, 0.0f);
            out[M] = make_float2(z.x - z.y, 0.0f);
        } else {
            float2 Xk, Xmk;
            rfft_pair(Z[k], Z[M - k], tw[k], Xk, Xmk);
            out[k] = Xk;
            out[M - k] = Xmk;
        }
    }
}

// N3  data.py:110-111  np.absolute(librosa.stft(y, n_fft)): centred frames (frame t covers
// y[t*H - N/2 : t*H + N/2]), edges reflected (numpy 'reflect': y[-i] = y[i], y[len-1+i] = y[len-1-i])
// or zero-filled, periodic Hann, magnitude only.  out: [B][T][M+1].
__global__ void __launch_bounds__(kThreads) k_analysis_mag(const float* __restrict__ y, long long y_stride,
                                                           long long len, float* __restrict__ out,
                                                           const float* __restrict__ win,
                                                           const float2* __restrict__ tw, int N, int H, int T,
                                                           int reflect) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* a = reinterpret_cast<float2*>(smem_raw);
    const int M = N >> 1;
    float2* b = a + M;
    const int t = blockIdx.x, bi = blockIdx.y;
    const float* yr = y + (long long)bi * y_stride;
    const long long s0 = (long long)t * H - M;
    for (int n = threadIdx.x; n < M; n += blockDim.x) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            long long s = s0 + 2 * n + e;
            bool inside = (s >= 0 && s < len);
            if (!inside && reflect) {
                s = s < 0 ? -s : 2 * (len - 1) - s;
                inside = (s >= 0 && s < len);
            }
            v[e] = inside ? yr[s] : 0.0f;
        }
        a[n] = make_float2(win[2 * n] * v[0], win[2 * n + 1] * v[1]);
    }
    __syncthreads();
    float2* Z = wg_fft<false>(a, b, M, N, tw);
    float* orow = out + ((long long)bi * T + t) * (M + 1);
    for (int k = threadIdx.x; k <= M / 2; k += blockDim.x) {
        if (k == 0) {
            float2 z = Z[0];
            orow[0] = fabsf(z.x + z.y);
            orow[M] = fabsf(z.x - z.y);
        } else {
            float2 Xk, Xmk;
            rfft_pair(Z[k], Z[M - k], tw[k], Xk, Xmk);
            orow[k] = sqrtf(fmaf(Xk.x, Xk.x, Xk.y * Xk.y));
            orow[M - k] = sqrtf(fmaf(Xmk.x, Xmk.x, Xmk.y * Xmk.y));
        }
    }
}

// shared tail: inverse FFT of Zin (in buffer `a`), window, scale 1/M, store frame
__device__ __forceinline__ void inverse_and_store(float2* a, float2* b, float* __restrict__ frame,
                                                  const float* __restrict__ win, const float2* __restrict__ tw,
                                                  int M, int N) {
    __syncthreads();
    float2* z = wg_fft<true>(a, b, M, N, tw);
    const float sc = 1.0f / (float)M;
    float2* f2 = reinterpret_cast<float2*>(frame);
    for (int n = threadIdx.x; n < M; n += blockDim.x) {
        float2 v = z[n];
        f2[n] = make_float2(win[2 * n] * (v.x * sc), win[2 * n + 1] * (v.y * sc));
    }
}

__global__ void __launch_bounds__(kThreads) k_generic_gl_frame(const float* __restrict__ x, long long x_stride,
                                                               const float* __restrict__ mag,
                                                               float* __restrict__ frames,
                                                               const float* __restrict__ win,
                                                               const float2* __restrict__ tw, int N, int H, int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* a = reinterpret_cast<float2*>(smem_raw);
    const int M = N >> 1;
    float2* b = a + M;
    const int t = blockIdx.x, bi = blockIdx.y;
    load_frame(a, x + (long long)bi * x_stride + (long long)t * H, win, M);
    float2* Z = wg_fft<false>(a, b, M, N, tw);
    float2* Zn = (Z == a) ? b : a;
    const float* mrow = mag + ((long long)bi * T + t) * (M + 1);
    for (int k = threadIdx.x; k <= M / 2; k += blockDim.x) {
        if (k == 0) {
            float2 z = Z[0];
            float2 X0 = gl_project(make_float2(z.x + z.y, 0.0f), mrow[0]);
            float2 XM = gl_project(make_float2(z.x - z.y, 0.0f), mrow[M]);
            Zn[0] = make_float2(0.5f * (X0.x + XM.x), 0.5f * (X0.x - XM.x));
        } else {
            float2 Xk, Xmk, zk, zm;
            float2 w = tw[k];
            rfft_pair(Z[k], Z[M - k], w, Xk, Xmk);
            Xk = gl_project(Xk, mrow[k]);
            Xmk = gl_project(Xmk, mrow[M - k]);
            irfft_pair(Xk, Xmk, w, zk, zm);
            Zn[k] = zk;
            Zn[M - k] = zm;
        }
    }
    inverse_and_store(Zn, Z, frames + ((long long)bi * T + t) * N, win, tw, M, N);
}

__global__ void __launch_bounds__(kThreads) k_generic_istft_frame(const float2* __restrict__ S,
                                                                  float* __restrict__ frames,
                                                                  const float* __restrict__ win,
                                                                  const float2* __restrict__ tw, int N, int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* a = reinterpret_cast<float2*>(smem_raw);
    const int M = N >> 1;
    float2* b = a + M;
    const int t = blockIdx.x, bi = blockIdx.y;
    const float2* in = S + ((long long)bi * T + t) * (M + 1);
    for (int k = threadIdx.x; k <= M / 2; k += blockDim.x) {
        if (k == 0) {
            float x0 = in[0].x, xm = in[M].x;  // imag(DC), imag(Nyquist) ignored like numpy irfft
            a[0] = make_float2(0.5f * (x0 + xm), 0.5f * (x0 - xm));
        } else {
            float2 zk, zm;
            irfft_pair(in[k], in[M - k], tw[k], zk, zm);
            a[k] = zk;
            a[M - k] = zm;
        }
    }
    inverse_and_store(a, b, frames + ((long long)bi * T + t) * N, win, tw, M, N);
}

// Gather-form overlap-add: y[s] = sum over frames n covering s, in frame order
// (the reference's `x[i:i+N] += ...` loop, audio_utilities.py:112-113).  No atomics.
__global__ void __launch_bounds__(kThreads) k_generic_ola(const float* __restrict__ frames, float* __restrict__ y,
                                                          int N, int H, int T, long long L) {
    const int bi = blockIdx.y;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    long long n_hi = s / H;
    if (n_hi > T - 1) n_hi = T - 1;
    long long n_lo = (s - N + H) / H;  // ceil((s-N+1)/H) for s >= N-1
    if (s < N) n_lo = 0;
    const float* fb = frames + (long long)bi * T * N;
    float acc = 0.0f;
    for (long long n = n_lo; n <= n_hi; ++n) acc += fb[n * N + (s - n * H)];
    y[(long long)bi * L + s] = acc;
}

}  // namespace

// scratch of the any-N kernels: the frames of one iteration and the ping-pong signals (gl_plan_create for
// geometries the fused kernels do not cover completely, gl_plan_set_path(1) otherwise)
int gl_generic_alloc(gl_plan* p) {
    int rc = 0;
    if (!p->d_frames) rc = gl_plan_alloc(p, (void**)&p->d_frames, (size_t)p->cap_frames * p->n_fft * sizeof(float));
    const size_t bytes = gl_signal_floats(p) * sizeof(float);
    if (!rc && !p->d_xa) rc = gl_plan_alloc(p, (void**)&p->d_xa, bytes);
    if (!rc && !p->d_xb) rc = gl_plan_alloc(p, (void**)&p->d_xb, bytes);
    return rc;
}

int gl_analysis_mag(gl_plan* p, const float* y, long long y_stride, long long len, float* mag, int reflect, int B,
                    int T, hipStream_t st) {
    const int N = p->n_fft;
    // n_fft 2048 on the fused path: one wave per frame on the register FFT core (gl_fast.hip)
    if (p->path == 0 && N == 2048 && gl_fast_supported(p, GL_FAST_OP_STFT))
        return gl_fast_analysis_mag(p, y, y_stride, len, mag, reflect, B, T, p->d_window_p, st);
    size_t lds = (size_t)N * sizeof(float2);
    hipLaunchKernelGGL(k_analysis_mag, dim3(T, B), dim3(kThreads), lds, st, y, y_stride, len, mag, p->d_window_p,
                       p->d_tw, N, p->hop, T, reflect);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_generic_stft(gl_plan* p, const float* x, long long x_stride, float* S, int B, int T, hipStream_t st) {
    const int N = p->n_fft;
    size_t lds = (size_t)N * sizeof(float2);  // 2 * M complex
    hipLaunchKernelGGL(k_generic_stft, dim3(T, B), dim3(kThreads), lds, st, x, x_stride, (float2*)S, p->d_window,
                       p->d_tw, N, p->hop, T);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

static int launch_ola(gl_plan* p, float* y, int B, int T, hipStream_t st) {
    const long long L = gl_len(p->n_fft, p->hop, T);
    dim3 grid((unsigned)((L + kThreads - 1) / kThreads), B);
    hipLaunchKernelGGL(k_generic_ola, grid, dim3(kThreads), 0, st, p->d_frames, y, p->n_fft, p->hop, T, L);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_generic_istft(gl_plan* p, const float* S, float* y, int B, int T, hipStream_t st) {
    if (!p->d_frames) return gl_set_error(GL_ERR_INVALID, "generic scratch absent: call gl_plan_set_path(plan, 1) first");
    const int N = p->n_fft;
    size_t lds = (size_t)N * sizeof(float2);
    hipLaunchKernelGGL(k_generic_istft_frame, dim3(T, B), dim3(kThreads), lds, st, (const float2*)S, p->d_frames,
                       p->d_window, p->d_tw, N, T);
    GL_HIP_CHECK(hipGetLastError());
    return launch_ola(p, y, B, T, st);
}

int gl_generic_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                   hipStream_t st) {
    if (!p->d_frames) return gl_set_error(GL_ERR_INVALID, "generic scratch absent: call gl_plan_set_path(plan, 1) first");
    int rc = 0;
    const int N = p->n_fft;
    const long long L = gl_len(N, p->hop, T);
    size_t lds = (size_t)N * sizeof(float2);
    const float* src = x0;
    const bool post = gl_post_needed(p);
    for (int it = 0; it < iters; ++it) {
        float* dst = (it == iters - 1) ? x_out : ((it & 1) ? p->d_xb : p->d_xa);
        hipLaunchKernelGGL(k_generic_gl_frame, dim3(T, B), dim3(kThreads), lds, st, src, L, mag, p->d_frames,
                           p->d_window, p->d_tw, N, p->hop, T);
        GL_HIP_CHECK(hipGetLastError());
        rc = launch_ola(p, dst, B, T, st);
        if (!rc && post) rc = gl_post_iter(p, dst, nullptr, dst, B, T, 1, 1, it == 0, it == iters - 1, st);
        if (rc) return rc;
        src = dst;
    }
    return 0;
}
