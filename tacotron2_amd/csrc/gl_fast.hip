// Fused wave-per-frame Griffin-Lim kernels (placeholder until the register-FFT
// kernels land; gl_fast_supported() gates every call).
#include "gl_internal.h"

bool gl_fast_supported(const gl_plan* p) {
    (void)p;
    return false;
}
int gl_fast_stft(gl_plan*, const float*, long long, float*, int, int, hipStream_t) {
    return gl_set_error(GL_ERR_UNSUPPORTED, "fused path not available for this geometry");
}
int gl_fast_istft(gl_plan*, const float*, float*, int, int, hipStream_t) {
    return gl_set_error(GL_ERR_UNSUPPORTED, "fused path not available for this geometry");
}
int gl_fast_run(gl_plan*, const float*, const float*, float*, int, int, int, hipStream_t) {
    return gl_set_error(GL_ERR_UNSUPPORTED, "fused path not available for this geometry");
}
