// Host-side helpers shared by the fused translation units (gl_fast.hip, gl_fast2048.hip, gl_fast1024.hip).
#pragma once
#include "gl_internal.h"

// frames per wave-run: the plan's pinned value, else chosen from (B, T) to fill the GPU in whole rounds
int gl_fast_pick_run_length(const gl_plan* p, int B, int T, int waves_per_cu);
bool gl_fast_regring_hop(const gl_plan* p);
long long gl_fast_small_form_limit(const gl_plan* p);
// x_out = x_main + run tails (after the last iteration of the kernels that keep tails)
int gl_fast_fold(gl_plan* p, const float* x_main, const float* tail, float* x_out, int B, int T, int G, int nruns,
                 hipStream_t st);
// x = overlap-add of the windowed inverse frames y [B][T][N] (the per-frame forms' final gather)
int gl_fast_ola_frames(const float* y, float* x, int B, int T, int N, int H, hipStream_t st);
// the Griffin-Lim loop per geometry
int gl_fast2048_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T, hipStream_t st);
int gl_fast1024_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T, hipStream_t st);
