/*
 * gl_hip.h  --  C-ABI of libgl_hip.so: MI355X (gfx950) Griffin-Lim vocoder kernels.
 *
 * Drop-in boundary for the reference's griffin_lim.py / audio_utilities.py path
 * (AidenMoosa/Tacotron2).  The reference is pure Python/numpy, so the "FFI" a
 * maintainer binds is ctypes; every entry point below names
 * the reference function (file:line) whose numpy calls it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and integer sizes only; no torch / HIP types.
 *   - All data pointers are DEVICE pointers to fp32 unless a comment says host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  The data-path
 *     calls (gl_run, gl_stft, gl_istft, gl_mel_to_mag, ...) only enqueue kernels / async
 *     memsets and copies on `stream` and return: no allocation, no synchronisation, no
 *     blocking copy -- legal inside a stream capture.  Memory is allocated and tables are
 *     uploaded (blocking) only by the configuration calls gl_plan_create[_capacity], gl_plan_set_path
 *     and gl_plan_set_options.
 *   - Every function returns 0 on success or a negative gl_status; the text of the
 *     last failure on the calling thread is returned by gl_last_error().
 *   - Geometry: N = n_fft (power of two, 64..4096), H = hop (>=1), T = frames,
 *     F = N/2+1 bins, L = T*H + N samples   (audio_utilities.py:109-110,173-174).
 *   - A plan is bound to one device and one (N, H); it owns the window, twiddle
 *     tables and scratch.  One plan must not be used from two threads at once, and calls
 *     on one plan must be ordered (one stream, or events between streams): they share
 *     the plan's scratch.
 *   - The library reads no environment variables.
 */
#ifndef GL_HIP_H_
#define GL_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gl_plan gl_plan;
typedef struct gl_graph gl_graph;

enum gl_status {
    GL_OK = 0,
    GL_ERR_INVALID = -1,   /* bad argument (geometry, null pointer, size beyond plan) */
    GL_ERR_HIP = -2,       /* a HIP runtime call failed; see gl_last_error()          */
    GL_ERR_NOMEM = -3,     /* device allocation failed                                */
    GL_ERR_UNSUPPORTED = -4
};

/* Library/ABI version (major*100 + minor). */
int gl_version(void);

/* Text of the last error raised on this thread ("" if none). */
const char* gl_last_error(void);

/* L = T*H + N                                     audio_utilities.py:110,174 */
long long gl_signal_length(int n_fft, int hop, int frames);

/* len(range(0, len - N, H)): frames the reference STFT produces for a signal of
 * `len` samples (0 when len <= N)                  audio_utilities.py:91-92   */
int gl_num_frames(int n_fft, int hop, long long len);

/* Create / destroy a plan.  max_frames / max_batch bound T and B of later calls; ALL scratch and
 * tables the plan's entry points can touch are allocated here (signals, run tails, packed
 * magnitudes, frame buffers -- about 14 KB per frame at n_fft 2048: 0.8 GB for 64 x 861 frames).
 * Window = symmetric Hann, np.hanning(N) (audio_utilities.py:88,108), computed in double on the
 * host, stored fp32.  gl_plan_create sizes the scratch for max_frames * max_batch frames.
 * gl_plan_create_capacity sizes it for max_total_frames instead (0 = max_frames * max_batch): calls
 * then need B <= max_batch, T <= max_frames AND B * T <= max_total_frames -- one plan serves a long
 * single utterance and a wide batch of short ones without holding memory for their product.
 * No counterpart in the reference (numpy allocates per call, audio_utilities.py:110,175).          */
int gl_plan_create(int n_fft, int hop, int max_frames, int max_batch, int device, gl_plan** out);
int gl_plan_create_capacity(int n_fft, int hop, int max_frames, int max_batch, long long max_total_frames,
                            int device, gl_plan** out);
int gl_plan_destroy(gl_plan* plan);

/* Bytes of device memory the plan currently holds (tables + scratch). */
long long gl_plan_device_bytes(const gl_plan* plan);

/* a-4  stft_for_reconstruction(x, fft_size, hopsamp)      audio_utilities.py:77-92
 *   x : [B][x_stride] fp32, each row needs >= (T-1)*H + N valid samples
 *   S : [B][T][F][2] fp32 (re, im interleaved == complex64), rows = time slices */
int gl_stft(gl_plan* plan, const float* x, long long x_stride, float* S, int B, int T, void* stream);

/* a-5  istft_for_reconstruction(X, fft_size, hopsamp)    audio_utilities.py:95-114
 *   S : [B][T][F][2] fp32 complex;  y : [B][L] fp32, L = T*H+N.
 *   Overlap-add of window*irfft(frame) in frame order, NO window-sum
 *   normalisation; imag(DC) and imag(Nyquist) are ignored as numpy's irfft does;
 *   the last H samples of every row are written as 0.                          */
int gl_istft(gl_plan* plan, const float* S, float* y, int B, int T, void* stream);

/* a-2 (first line)  np.dot(filterbank.T, mel).T              griffin_lim.py:18-19
 *   filterbank : [n_mels][F] fp32 (the array librosa.filters.mel returns,
 *                griffin_lim.py:13-14; any basis of that shape may be passed)
 *   mel        : [B][n_mels][T] fp32;    mag : [B][T][F] fp32
 *   fp32-input MFMA (v_mfma_f32_32x32x2_f32), exact fp32 fma chain over n_mels.   */
int gl_mel_to_mag(gl_plan* plan, const float* filterbank, const float* mel, float* mag,
                  int n_mels, int B, int T, void* stream);

/* N1  infer.py:24-29 glue fused into a-2's contraction: the decoder's output goes straight in.
 *   logmel : [B][T][n_mels] fp32, natural-log mel as Tacotron2.inference returns it (y_pred_post);
 *   computes dynamic_range_decompression (data.py:84-85: exp(mel) / compression_factor), the
 *   transpose of infer.py:28 and np.dot(filterbank.T, mel).T (griffin_lim.py:18-19) in one kernel:
 *   mag[b][t][f] = sum_m exp(logmel[b][t][m]) / compression_factor * filterbank[m][f]            */
int gl_logmel_to_mag(gl_plan* plan, const float* filterbank, const float* logmel, float* mag,
                     int n_mels, float compression_factor, int B, int T, void* stream);

/* N3  the forward analysis path, data.py:100-116 (audiopath_to_mel after the file is loaded):
 *   stft = librosa.stft(y, n_fft)  -- hop = the plan's hop (librosa's default is n_fft/4), periodic Hann,
 *   center=True: frame t covers y[t*H - N/2, t*H + N/2), T = 1 + len / H frames (gl_analysis_num_frames);
 *   pad_mode 1 = edges reflected (librosa < 0.10, the reference's era), 0 = zero-filled (librosa >= 0.10);
 *   mag = |stft| (data.py:111), mel = filterbank @ mag on MFMA (data.py:112),
 *   logmel = log(max(mel, clip_min) * compression_factor) (data.py:80-81, clip_min 0.01, factor 1).
 *   y : [B][y_stride] fp32 (first `len` samples);  mag : [B][T][F] fp32 out (also the scratch of the
 *   contraction);  logmel : [B][n_mels][T] fp32 out -- the (80, T) layout the reference's mels have.
 *   librosa is not available to pin this against: parity unpinned, checked against a numpy
 *   statement of librosa's documented algorithm.                                                       */
int gl_analysis_num_frames(int hop, long long len);
int gl_wav_to_logmel(gl_plan* plan, const float* y, long long y_stride, long long len, const float* filterbank,
                     float* mag, float* logmel, int n_mels, int pad_mode, float clip_min, float compression_factor,
                     int B, void* stream);

/* N3  the sample-rate conversion inside librosa.load(path), data.py:101 (default sr = 22 050 Hz, res_type
 *   'kaiser_best' in the librosa 0.6 / 0.7 of the reference's Python 3.7 environment = resampy 0.2's
 *   band-limited sinc interpolation: 64 zero crossings, 512 table points per crossing read with linear
 *   interpolation, Kaiser window beta 14.769656459379492, roll-off 0.9475937167399596, filter scaled by the
 *   ratio when downsampling).  y has gl_resample_length = ceil(n_in * sr_out / sr_in) samples like
 *   librosa.resample (the last one is zero where int(n_in * ratio) is smaller: librosa.util.fix_length).
 *   x : [B][x_stride] fp32 (n_in valid samples each);  y : [B][y_stride] fp32.  Neither librosa nor resampy
 *   is pinned by the reference or available here: parity unpinned, checked against a numpy statement of the
 *   algorithm and, for band-limited signals, against scipy's polyphase resampler.  The reference's own data
 *   (LJSpeech) is already at 22 050 Hz, so its pipeline never resamples.                                  */
long long gl_resample_length(long long n_in, int sr_in, int sr_out);
int gl_resample(gl_plan* plan, const float* x, long long x_stride, long long n_in, float* y, long long y_stride,
                int sr_in, int sr_out, int B, void* stream);

/* a-3 + a-6  reconstruct_signal_griffin_lim(mag, fft_size, hopsamp, iterations)
 *                                                     audio_utilities.py:154-188
 *   mag   : [B][T][F] fp32   (rows = time slices)
 *   x0    : [B][L] fp32 start signal (the reference draws np.random.randn(L),
 *           audio_utilities.py:176; the host wrapper draws it the same way or the
 *           caller fills it with gl_randn)
 *   x_out : [B][L] fp32; may alias x0.
 *   Runs `iters` times  x <- ISTFT( mag * S/|S| ),  S = STFT(x), with |S|==0 ->
 *   phase 1+0j (np.angle(0)==0, audio_utilities.py:181-183), entirely on
 *   `stream` with no host synchronisation.  The dead RMSE (line 186) is not
 *   computed.  iters == 0 copies x0 to x_out.                                  */
int gl_run(gl_plan* plan, const float* mag, const float* x0, float* x_out,
           int iters, int B, int T, void* stream);

/* a-2 + a-3 in one call                                   griffin_lim.py:18-24
 *   np.dot(filterbank.T, mel).T followed by reconstruct_signal_griffin_lim, without the [B][T][F] magnitude
 *   array in between: the mel inversion stores the magnitudes straight in the per-lane order the loop's
 *   kernels read (plan-owned), so the repacking pass gl_run starts with and one round trip of the magnitudes
 *   through HBM are gone.  For callers that do not need the linear magnitude save_mel_to_wav returns
 *   (griffin_lim.py:35) -- batched serving.  Results equal gl_mel_to_mag (+ gl_mask_frames) + gl_run bit for bit.
 *   mel_layout 0 : mel [B][n_mels][T] linear amplitude (as gl_mel_to_mag; compression_factor ignored)
 *   mel_layout 1 : mel [B][T][n_mels] natural-log amplitude (as gl_logmel_to_mag)
 *   frames       : NULL, or [B] int32 on the device: frames t >= frames[b] get zero magnitude (gl_mask_frames)
 *   x0, x_out, iters : as gl_run.
 *   Exists where the loop reads packed magnitudes: n_fft 2048, hop 128/256/512/1024, default path, x0 and
 *   x_out 8-byte aligned (gl_mel_run_supported tells for the plan); elsewhere GL_ERR_UNSUPPORTED and the
 *   caller makes the two calls.                                                                              */
int gl_mel_run_supported(const gl_plan* plan);
int gl_mel_run(gl_plan* plan, const float* filterbank, const float* mel, int n_mels, int mel_layout,
               float compression_factor, const int* frames, const float* x0, float* x_out, int iters,
               int B, int T, void* stream);

/* Ragged batches.  The reference reconstructs one utterance per call (audio_utilities.py:154), so
 * utterances of different lengths never meet; a batch here has one frame count T for all.  Zeroing
 * the magnitude rows t >= frames[b] makes the padding frames project to exactly zero (they add
 * nothing to the overlap-add), so gl_run on the padded batch gives utterance b the samples
 * [0, frames[b]*H + N) it would get alone and zeros after them.  frames : [B] int32 on the device.   */
int gl_mask_frames(gl_plan* plan, float* mag, const int* frames, int B, int T, void* stream);

/* Select the code path gl_run / gl_stft / gl_istft use: 0 = auto (fused
 * wave-per-frame kernels where the geometry has one, else generic),
 * 1 = force the generic (any power-of-two N) kernels.  For tests/benchmarks.
 * A configuration call: the first switch to 1 on a geometry with fused kernels
 * allocates the generic kernels' scratch (not for use inside a stream capture). */
int gl_plan_set_path(gl_plan* plan, int path);

/* Off-by-default variants the reference does not have (north_star names the first two):
 *   GL_OPT_WSUM_NORMALISE  every inverse STFT -- inside gl_run and in gl_istft -- divides sample s by
 *                          sum_t w^2(s - tH) (least-squares ISTFT of Griffin & Lim 1984; the reference's
 *                          audio_utilities.py:108-114 leaves the sum in, gain ~ 0.375 N/H).  Samples no
 *                          frame reaches are left as they are.
 *   GL_OPT_CLAMP_MAG       gl_mel_to_mag / gl_logmel_to_mag write max(0, .): a pseudo-inverse basis
 *                          (instead of griffin_lim.py:18's filterbank.T) produces negative magnitudes.
 *   momentum in [0, 1)     fast Griffin-Lim (Perraudin, Balazs, Sondergaard 2013): the next iteration
 *                          starts from x_n + momentum * (x_n - x_{n-1}); 0 = the reference's plain loop.
 * With flags == 0 and momentum == 0 (the default) every entry point computes exactly what the reference
 * does.  The variants cost one extra streaming pass over x per iteration.  A configuration call: the
 * first time a variant is switched on its buffers are allocated (previous iterate; window-sum table,
 * which the data-path calls then (re)build for their frame count with a small kernel on their stream: only
 * when the frame count changes -- until a stream capture has recorded a call with GL_OPT_WSUM_NORMALISE on the
 * plan; from then on every call rebuilds it, since graph replays rewrite the table unseen by the host).   */
enum { GL_OPT_WSUM_NORMALISE = 1, GL_OPT_CLAMP_MAG = 2 };
int gl_plan_set_options(gl_plan* plan, int flags, float momentum);

/* Frames each wavefront of the fused kernels processes in one run (0 = automatic: chosen from
 * B, T and the CU count to fill the GPU in whole rounds).  Results are bit-reproducible for a
 * given run length; overlap-add sums at run seams associate differently for different run
 * lengths (last-bit differences), so pin it when outputs of different batch sizes -- e.g.
 * per-GPU shards vs. one big batch -- must be bit-identical.  Values below ceil((N-H)/H) are
 * raised to it.  A pinned value also keeps gl_run on the run kernels: unpinned, small problems
 * (a few thousand frames) take a per-frame form whose results agree with them to rounding only.
 * No counterpart in the reference (a property of the GPU work split).                          */
int gl_plan_set_run_frames(gl_plan* plan, int frames);

/* north_star (d): "the full N-iteration loop kept on-device on one HIP stream, and optionally
 * hipGraph-captured".  The reference has no counterpart (its loop is a Python while,
 * audio_utilities.py:178).  gl_plan_create / gl_plan_set_path / gl_plan_set_options allocate and upload
 * everything a plan can touch, so every data-path call is legal inside a stream capture from the start (the
 * gl_plan_prepare of ABI 1.x / 2.x, a no-op since 2.0, is gone in 3.0).  gl_graph_begin starts capturing `stream` (non-default,
 * relaxed mode); every gl_* call enqueued on it until gl_graph_end becomes a node of the
 * graph instead of running; gl_graph_end instantiates it.  gl_graph_launch replays the whole
 * sequence (e.g. gl_mel_to_mag + gl_run(60) + gl_peak_normalise = 64 kernels) with one submission,
 * on the same buffers it was captured with.  Plan settings (path, run length) are frozen in.       */
int gl_graph_begin(void* stream);
int gl_graph_end(void* stream, gl_graph** out);
int gl_graph_launch(gl_graph* graph, void* stream);
int gl_graph_destroy(gl_graph* graph);

/* Device standard-normal fill (Philox4x32-10 + Box-Muller), for throughput runs
 * where bit-parity with numpy's MT19937 stream (audio_utilities.py:176) is not
 * needed.  out : [n] fp32.  Deterministic in (seed, index).                     */
int gl_randn(float* out, long long n, uint64_t seed, void* stream);

/* As gl_randn, but element i of the call is element offset+i of the seed's stream: ranks that
 * each hold a slice of one long start signal (config 4) fill it consistently.              */
int gl_randn_at(float* out, long long n, uint64_t seed, long long offset, void* stream);

/* Long-form stitching (BASELINE config 4; the reference reconstructs one utterance in
 * one piece, audio_utilities.py:154-188, so this has no line there): out[out_offset + i] +=
 * w(i) * chunk[i] for i in [0, chunk_len), samples falling outside [0, out_len) are skipped.
 * w ramps 0 -> 1 over the first fade_in samples and 1 -> 0 over the last fade_out samples
 * (fade_in + fade_out <= chunk_len; 0 = hard cut); two chunks whose fades overlap over the
 * same F samples sum to weight exactly 1.  `out` must be zero-filled before the first chunk
 * and the calls that overlap must be ordered on one stream.                                 */
int gl_stitch_add(float* out, long long out_len, const float* chunk, long long chunk_len, long long out_offset,
                  long long fade_in, long long fade_out, void* stream);

/* a-2 (normalise step)  max_sample = max|x|; if max_sample > 1: x /= max_sample
 *                                                        griffin_lim.py:27-29
 *   x : [B][row_stride] fp32, first `len` samples of each row are used, in place
 *   peaks : [B] fp32 out (max|x| per row before scaling); may be NULL
 *   mode : 0 = reference behaviour (scale only when peak > 1)
 *          1 = always scale to peak 1 (what save_audio_to_file then does anyway,
 *              audio_utilities.py:202-204); rows with peak 0 are left untouched  */
int gl_peak_normalise(gl_plan* plan, float* x, long long row_stride, long long len,
                      float* peaks, int mode, int B, void* stream);

/* No-allocation variant of gl_peak_normalise.  NO COUNTERPART IN THE REFERENCE (griffin_lim.py:27-29 is three
 * numpy lines): it exists for callers that cannot allocate device memory -- code under hipGraph / stream
 * capture, tight serving loops, harnesses that forbid hipMalloc -- and therefore cannot create a gl_plan
 * (gl_plan_create allocates).  Same arguments, kernels and results as gl_peak_normalise, bit for bit, but the
 * per-row maxima the two kernels exchange live in a file-scope device array of GL_STATIC_MAX_BATCH entries
 * (16 KB, held by every GPU the library runs on until the process ends) instead of plan-owned memory, and
 * `device` names the GPU the buffers are on instead of a plan.  The call makes no hipMalloc / hipFree or any
 * other allocating call; the first call on a device looks the array's address up once
 * (hipGetSymbolAddress).  B > GL_STATIC_MAX_BATCH is rejected with GL_ERR_INVALID.  All calls on one GPU
 * share the array (it keeps the last call's maxima; each call clears the B entries it uses first on
 * `stream`), so calls that overlap in time on one GPU -- different streams, or threads -- race: order them
 * on one stream.                                                                                        */
#define GL_STATIC_MAX_BATCH 4096
int gl_peak_normalise_static(float* x, long long row_stride, long long len, float* peaks, int mode, int B,
                             int device, void* stream);

/* N2 / a-7  sample conversion of save_audio_to_file            audio_utilities.py:202-210
 *   x_max = max|x|; x /= x_max; x *= 32767; int(round(x[i]))  -- in float64 with round-half-even like
 *   the reference, so the bytes match it for the same input samples.
 *   x : [B][row_stride] fp32 (first `len` samples used, not modified); pcm : [B][pcm_stride] int16
 *   peaks : [B] fp32 out (x_max per row) or NULL.  The reference asserts x_max <= 1 and fails on
 *   x_max == 0 (NaN): a device call cannot raise, so rows with x_max == 0 are written as zeros and
 *   the caller checks `peaks` (the Python wrapper raises like the reference).                        */
int gl_pcm16_encode(gl_plan* plan, const float* x, long long row_stride, long long len, int16_t* pcm,
                    long long pcm_stride, float* peaks, int B, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GL_HIP_H_ */
