// Host side shared by the fused kernels (gl_fast2048.hip, gl_fast1024.hip): constant tables, plan-owned
// scratch, the run-length choice, and the small streaming kernels around the loop (tail fold, per-frame
// overlap-add gather, optional post pass).
#include <math.h>
#include <string.h>

#include <vector>

#include "gl_fast_host.h"
#include "gl_fft1024.h"

using namespace gl_fft;

namespace {

// x[b][s] = sum over frames m (ascending) of y[b][m][s - mH]: the overlap-add of audio_utilities.py:111-113 as a
// gather; the last H samples no frame covers come out zero, like the reference's.
__global__ void __launch_bounds__(256) k_ola_frames(const float* __restrict__ y, float* __restrict__ x, long long L, int N,
                                                    int H, int T) {
    const int b = blockIdx.y;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    const float* yb = y + (long long)b * T * N;
    long long m_lo = (s - N + H) / H;             // ceil((s - N + 1) / H) for s - N + 1 > 0
    if (s - N + 1 <= 0) m_lo = 0;
    long long m_hi = s / H;
    if (m_hi > T - 1) m_hi = T - 1;
    float v = 0.0f;
    for (long long m = m_lo; m <= m_hi; ++m) v += yb[m * N + (s - m * H)];
    x[(long long)b * L + s] = v;
}
// x_out[s] = x_main[s] + tail of the previous run (final fold after the last iteration)
__global__ void __launch_bounds__(256) k_fold_tails(const float* __restrict__ x_main, const float* __restrict__ tail,
                                                    float* __restrict__ x_out, long long L, int g_h, int nruns,
                                                    int ovl) {
    const int b = blockIdx.y;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    x_out[(long long)b * L + s] =
        fetch_x(x_main + (long long)b * L, tail + (long long)b * nruns * ovl, s, g_h, nruns, ovl);
}

// Post pass of the optional (non-reference) variants, one launch per iteration when enabled:
// v = x_main[s] + tail; window-sum normalisation v *= inv_wsum[s]; fast Griffin-Lim momentum
// t = v + alpha (v - prev[s]), prev[s] = v (Perraudin et al. 2013, in the signal domain: the STFT
// is linear, so extrapolating the signals extrapolates the consistent spectrograms).  The first
// iteration has no predecessor and the last one returns v itself.
__global__ void __launch_bounds__(256) k_post_iter(const float* __restrict__ x_main, const float* __restrict__ tail,
                                                   const float* __restrict__ inv_wsum, float* __restrict__ prev,
                                                   float* __restrict__ out, long long L, int g_h, int nruns, int ovl,
                                                   float alpha, int first, int last) {
    const int b = blockIdx.y;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    float v = fetch_x(x_main + (long long)b * L, tail ? tail + (long long)b * nruns * ovl : nullptr, s, g_h, nruns, ovl);
    if (inv_wsum) v *= inv_wsum[s];
    float t = v;
    if (prev) {
        if (!first && !last) t = v + alpha * (v - prev[(long long)b * L + s]);
        if (!last) prev[(long long)b * L + s] = v;
    }
    out[(long long)b * L + s] = t;
}

// inv[s] = 1 / sum over the frames t < T covering s of w^2(s - tH), in double from the plan's w^2 table; samples no
// frame reaches, or where the sum underflows, get 1 (left unscaled: the convention of librosa's istft).
__global__ void __launch_bounds__(256) k_build_inv_wsum(const double* __restrict__ w2, float* __restrict__ inv, long long L,
                                                        int N, int H, int T) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    long long t_lo = (s - N + H) / H;             // ceil((s - N + 1) / H) for s - N + 1 > 0
    if (s - N + 1 <= 0) t_lo = 0;
    long long t_hi = s / H;
    if (t_hi > T - 1) t_hi = T - 1;
    double acc = 0.0;
    for (long long t = t_lo; t <= t_hi; ++t) acc += w2[s - t * H];
    inv[s] = acc > 1e-20 ? (float)(1.0 / acc) : 1.0f;
}

// per-lane constants of k_fast2048 (entry-major, tab[e*64 + lane])
int build_table1(gl_plan* p) {
    const int N = p->n_fft;
    std::vector<float> tab((size_t)kTabEntries * 64, 0.0f);
    const double pi = 3.14159265358979323846;
    auto W = [&](int n, long long k, float* re, float* im) {
        double a = -2.0 * pi * (double)(k % n) / (double)n;
        *re = (float)cos(a);
        *im = (float)sin(a);
    };
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 16; ++j)
            for (int e = 0; e < 2; ++e) {
                int n = 2 * (l + 64 * j) + e;
                tab[(size_t)(kTabWin + 2 * j + e) * 64 + l] = (float)(0.5 - 0.5 * cos(2.0 * pi * n / (double)(N - 1)));
            }
        for (int k2 = 1; k2 < 16; ++k2)
            W(1024, (long long)l * k2, &tab[(size_t)(kTabTw1 + 2 * (k2 - 1)) * 64 + l],
              &tab[(size_t)(kTabTw1 + 2 * (k2 - 1) + 1) * 64 + l]);
        for (int d = 1; d < 8; ++d)
            W(64, (long long)(l & 7) * d, &tab[(size_t)(kTabTw2 + 2 * (d - 1)) * 64 + l],
              &tab[(size_t)(kTabTw2 + 2 * (d - 1) + 1) * 64 + l]);
        for (int u = 0; u < 9; ++u) {
            int k = (u == 8) ? 512 : (l ? l + 128 * u : (u < 4 ? 128 * u : 64 + 128 * (u - 4)));
            W(2048, k, &tab[(size_t)(kTabTwp + 2 * u) * 64 + l], &tab[(size_t)(kTabTwp + 2 * u + 1) * 64 + l]);
        }
    }
    int rc = gl_plan_alloc(p, (void**)&p->d_fast_tab, tab.size() * sizeof(float));
    if (rc) return rc;
    GL_HIP_CHECK(hipMemcpy(p->d_fast_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// pair-layout table of the register-FFT kernels (gl_fft1024.h: kT2*, TE) + one row of zeros behind it
int build_table2(gl_plan* p) {
    const int N = p->n_fft;
    std::vector<float> tab((size_t)(kT2Zero + 1) * 64 * 2, 0.0f);
    const double pi = 3.14159265358979323846;
    auto put = [&](int e, int l, double re, double im) {      // pair layout, see TE()
        const size_t at = ((size_t)(e >> 1) * 128 + 2 * (size_t)l + (e & 1)) * 2;
        tab[at] = (float)re;
        tab[at + 1] = (float)im;
    };
    auto Wre = [&](int n, long long k) { return cos(-2.0 * pi * (double)(k % n) / (double)n); };
    auto Wim = [&](int n, long long k) { return sin(-2.0 * pi * (double)(k % n) / (double)n); };
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 16; ++j) {
            // n_fft 2048: lane holds the even/odd sample pair 2(l+64j), +1 of ONE frame;
            // n_fft 1024: lane holds sample l+64j of TWO frames (same window value twice)
            int n0 = (N == kN) ? 2 * (l + 64 * j) : (l + 64 * j);
            int n1 = (N == kN) ? n0 + 1 : n0;
            put(kT2Win + j, l, 0.5 - 0.5 * cos(2.0 * pi * n0 / (double)(N - 1)), 0.5 - 0.5 * cos(2.0 * pi * n1 / (double)(N - 1)));
        }
        for (int k2 = 1; k2 < 16; ++k2) put(kT2Tw1 + k2 - 1, l, Wre(1024, (long long)l * k2), Wim(1024, (long long)l * k2));
        for (int d = 1; d < 8; ++d) put(kT2Tw2 + d - 1, l, Wre(64, (long long)(l & 7) * d), Wim(64, (long long)(l & 7) * d));
        for (int u = 0; u < 9; ++u) {
            int k = (u == 8) ? 512 : (l ? l + 128 * u : (u < 4 ? 128 * u : 64 + 128 * (u - 4)));
            put(kT2Twp + u, l, Wim(2048, k), -Wre(2048, k));   // -i * W2048^k (middle_project)
        }
    }
    int rc = gl_plan_alloc(p, (void**)&p->d_fast_tab2, tab.size() * sizeof(float));
    if (rc) return rc;
    GL_HIP_CHECK(hipMemcpy(p->d_fast_tab2, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

// Frames per wave-run.  A run costs its frames plus a fixed prologue/epilogue (ring fill,
// tail write, ~1.5 frame-times); the launch finishes after ceil(runs / resident waves)
// rounds, so choose the G that minimises rounds * (G + overhead).  G*H >= N-H keeps the
// number of runs that touch one sample at two.
int gl_fast_pick_run_length(const gl_plan* p, int B, int T, int waves_per_cu) {
    const int N = p->n_fft, H = p->hop;
    int g_min = (N - H + H - 1) / H;
    if (g_min < 1) g_min = 1;
    if (p->run_frames > 0) {
        int G = p->run_frames < g_min ? g_min : p->run_frames;
        return G > T ? T : G;
    }
    if (T <= g_min) return T;
    const long long slots = (long long)p->n_cu * waves_per_cu;
    int best = g_min;
    double best_cost = 1e300;
    const int g_max = T < 96 ? T : 96;
    for (int G = g_min; G <= g_max; ++G) {
        const long long runs = (long long)B * ((T + G - 1) / G);
        const long long rounds = (runs + slots - 1) / slots;
        const double cost = (double)rounds * (G + 1.5);
        if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && G > best)) {
            best_cost = cost;
            best = G;
        }
    }
    return best;
}

bool gl_fast_supported(const gl_plan* p, int op) {
    if (p->hop < 1) return false;
    if (p->n_fft == kN && p->hop <= kN) return true;                          // all three entry points
    if (p->n_fft == 1024 && p->hop <= 1024) return op == GL_FAST_OP_RUN;       // Griffin-Lim loop only
    return false;
}

// n_fft 2048 with a hop of 1, 2, 4 or 8 register rows (128 samples): the register-ring kernels apply
bool gl_fast_regring_hop(const gl_plan* p) {
    if (p->n_fft != kN || p->hop % 128 != 0) return false;
    const int hs = p->hop / 128;
    return hs == 1 || hs == 2 || hs == 4 || hs == 8;
}

// Largest B*T the per-frame ("small problem") forms serve, 0 where the geometry has none: n_fft 2048 at hops of
// 256, 512, 1024 up to 8 frames per CU; n_fft 1024 with reach floor(1023/H) in [1, 8] up to 20 frames per CU
// (the measurements behind the two factors are quoted at the call sites).
long long gl_fast_small_form_limit(const gl_plan* p) {
    if (p->n_fft == kN) return (gl_fast_regring_hop(p) && p->hop >= 256 && p->hop < kN) ? 8LL * p->n_cu : 0;
    if (p->n_fft == 1024) {
        const int reach = 1023 / p->hop;
        return (reach >= 1 && reach <= 8) ? 20LL * p->n_cu : 0;
    }
    return 0;
}

// Everything the fused entry points of this plan can touch, allocated and uploaded once at gl_plan_create, so
// that gl_run / gl_stft / gl_istft only enqueue kernels afterwards (no allocation, no synchronisation; legal
// inside a stream capture).
int gl_fast_alloc(gl_plan* p) {
    const int N = p->n_fft, H = p->hop;
    const size_t xbytes = gl_signal_floats(p) * sizeof(float);
    int rc = gl_plan_alloc(p, (void**)&p->d_xa, xbytes);
    if (!rc) rc = gl_plan_alloc(p, (void**)&p->d_xb, xbytes);
    if (!rc) rc = build_table2(p);
    if (!rc && N == 1024) rc = gl_plan_alloc(p, (void**)&p->d_xc, xbytes);      // read / write / clear rotation
    if (!rc && N == kN) {
        // run tails, [B][runs][N-H] dense: a run has at least g_min frames unless it is its utterance's only or last
        // one, so a call has at most B * T / g_min + B <= cap_frames / g_min + max_B runs
        int g_min = (N - H + H - 1) / H;
        if (g_min < 1) g_min = 1;
        const size_t max_runs = (size_t)(p->cap_frames / g_min) + (size_t)p->max_B + 1;
        const size_t tbytes = max_runs * (size_t)(N - H) * sizeof(float);
        rc = gl_plan_alloc(p, (void**)&p->d_tail_a, tbytes);
        if (!rc) rc = gl_plan_alloc(p, (void**)&p->d_tail_b, tbytes);
        if (!rc) rc = build_table1(p);
        if (!rc && gl_fast_regring_hop(p))
            rc = gl_plan_alloc(p, (void**)&p->d_magp, (size_t)p->cap_frames * kMagPack * sizeof(float));
    }
    long long fr = p->cap_frames;
    if (fr > gl_fast_small_form_limit(p)) fr = gl_fast_small_form_limit(p);
    if (!rc && fr > 0) {                                   // two sets of windowed inverse frames
        p->yframes_bytes = (size_t)2 * fr * N * sizeof(float);
        rc = gl_plan_alloc(p, (void**)&p->d_yframes, p->yframes_bytes);
    }
    return rc;
}

int gl_fast_fold(gl_plan* p, const float* x_main, const float* tail, float* x_out, int B, int T, int G, int nruns,
                 hipStream_t st) {
    const long long L = gl_len(p->n_fft, p->hop, T);
    dim3 grid((unsigned)((L + 255) / 256), B);
    hipLaunchKernelGGL(k_fold_tails, grid, dim3(256), 0, st, x_main, tail, x_out, L, G * p->hop, nruns,
                       p->n_fft - p->hop);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_fast_ola_frames(const float* y, float* x, int B, int T, int N, int H, hipStream_t st) {
    const long long L = gl_len(N, H, T);
    hipLaunchKernelGGL(k_ola_frames, dim3((unsigned)((L + 255) / 256), B), dim3(256), 0, st, y, x, L, N, H, T);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_post_iter(gl_plan* p, const float* x_main, const float* tail, float* out, int B, int T, int G, int nruns,
                 bool first, bool last, hipStream_t st) {
    const long long L = gl_len(p->n_fft, p->hop, T);
    dim3 grid((unsigned)((L + 255) / 256), B);
    hipLaunchKernelGGL(k_post_iter, grid, dim3(256), 0, st, x_main, tail,
                       (p->opt_flags & GL_OPT_WSUM_NORMALISE) ? p->d_inv_wsum : nullptr,
                       p->momentum != 0.0f ? p->d_prev : nullptr, out, L, G * p->hop, nruns, p->n_fft - p->hop,
                       p->momentum, first ? 1 : 0, last ? 1 : 0);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

// The window-sum table of the normalised inverse STFT depends on the frame count of the call: (re)built on the
// call's own stream by a small kernel whenever T differs from the table's.  Inside a stream capture the build is
// always recorded (it then belongs to the graph), and from then on EVERY call enqueues it: a replay of that graph
// rewrites the table for the captured T without the host seeing it, so the cached frame count can no longer be
// trusted (capture T1 / eager T2 / replay / eager T2 would otherwise normalise with T1's sums).
int gl_wsum_enqueue(gl_plan* p, int T, hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st) (void)hipStreamIsCapturing(st, &cs);
    const bool capturing = cs != hipStreamCaptureStatusNone;
    if (capturing) p->wsum_captured = 1;
    if (!p->wsum_captured && p->wsum_T == T) return 0;
    const long long L = gl_len(p->n_fft, p->hop, T);
    hipLaunchKernelGGL(k_build_inv_wsum, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, p->d_w2, p->d_inv_wsum, L,
                       p->n_fft, p->hop, T);
    GL_HIP_CHECK(hipGetLastError());
    p->wsum_T = T;
    return 0;
}

int gl_fast_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                hipStream_t st) {
    return p->n_fft == 1024 ? gl_fast1024_run(p, mag, x0, x_out, iters, B, T, st)
                            : gl_fast2048_run(p, mag, x0, x_out, iters, B, T, st);
}
