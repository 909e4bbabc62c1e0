"""Plan objects: one (n_fft, hop, device) geometry bound to libgl_hip scratch.

PyTorch-ROCm is used only for device memory and streams; all compute goes
through the C-ABI in include/gl_hip.h.
"""
import contextlib
import ctypes
import threading

import torch

from . import _lib


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise RuntimeError("tacotron2_amd needs a ROCm GPU (MI355X / gfx950): torch.cuda.is_available() is False "
                           "and there is no CPU fallback for the Griffin-Lim path")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise ValueError(f"device must be a cuda (ROCm) device, got {dev}")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class Plan:
    """Owns a gl_plan handle.  Methods take/return fp32 torch tensors on the plan's device."""

    def __init__(self, n_fft, hop, max_frames, max_batch, device=None, max_total_frames=None):
        """max_total_frames: frames (B * T) the plan's scratch is sized for; None = max_frames * max_batch.  A call
        needs T <= max_frames, B <= max_batch and B * T <= max_total_frames (gl_plan_create_capacity)."""
        self.device = _require_gpu(device)
        self.n_fft = int(n_fft)
        self.hop = int(hop)
        self.max_frames = int(max_frames)
        self.max_batch = int(max_batch)
        self.max_total_frames = self.max_frames * self.max_batch if max_total_frames is None else int(max_total_frames)
        self.n_bins = self.n_fft // 2 + 1
        self._h = ctypes.c_void_p()
        self._lib = _lib.lib()
        self._path = 0            # plan defaults; Plan.run(path=..., run_frames=...) overrides them for ONE call
        self._run_frames = 0
        self._shared = False      # True for plans handed out by get_plan(): their defaults are frozen
        _lib.check(self._lib.gl_plan_create_capacity(self.n_fft, self.hop, self.max_frames, self.max_batch,
                                                     self.max_total_frames, self.device.index,
                                                     ctypes.byref(self._h)), "gl_plan_create_capacity")

    # -- lifetime ---------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.gl_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------------
    def signal_length(self, frames):
        return int(frames) * self.hop + self.n_fft

    def device_bytes(self):
        return int(self._lib.gl_plan_device_bytes(self._h))

    def set_path(self, path):
        """Plan default: 0 = auto (fused kernels where available), 1 = force the generic kernels.
        Plans from get_plan() are shared between callers: pass path= to the call instead."""
        self._not_shared("set_path")
        self._set_path(path)

    def set_run_frames(self, frames):
        """Plan default for the frames-per-wave-run of the fused kernels (0 = automatic).  Pin it when
        results for different batch sizes (per-GPU shards vs one batch) must be bit-identical.
        Plans from get_plan() are shared between callers: pass run_frames= to the call instead."""
        self._not_shared("set_run_frames")
        self._set_run_frames(frames)

    def _not_shared(self, what):
        if self._shared:
            raise RuntimeError(f"{what}() on a plan from get_plan(): the cache hands the same plan to every caller "
                               "of this geometry, so its defaults stay as they are -- pass path= / run_frames= to "
                               "the call (they apply to that call only) or create a private Plan")

    def _set_path(self, path):
        _lib.check(self._lib.gl_plan_set_path(self._h, int(path)), "gl_plan_set_path")
        self._path = int(path)

    def _set_run_frames(self, frames):
        _lib.check(self._lib.gl_plan_set_run_frames(self._h, int(frames)), "gl_plan_set_run_frames")
        self._run_frames = int(frames)

    @contextlib.contextmanager
    def _scoped(self, path=None, run_frames=None):
        """path / run_frames for the calls inside the block only; the plan's own values come back afterwards,
        also when a call raises."""
        old = (self._path, self._run_frames)
        try:
            if path is not None and int(path) != old[0]:
                self._set_path(path)
            if run_frames is not None and int(run_frames) != old[1]:
                self._set_run_frames(run_frames)
            yield self
        finally:
            if self._path != old[0]:
                self._set_path(old[0])
            if self._run_frames != old[1]:
                self._set_run_frames(old[1])

    def set_options(self, window_sum_normalize=False, clamp_mag=False, momentum=0.0):
        """Off-by-default variants (gl_plan_set_options); the defaults are the reference's behaviour."""
        flags = (1 if window_sum_normalize else 0) | (2 if clamp_mag else 0)
        _lib.check(self._lib.gl_plan_set_options(self._h, flags, float(momentum)), "gl_plan_set_options")

    def _f32(self, t, shape=None):
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("expected a contiguous float32 tensor on %s" % (self.device,))
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return ctypes.c_void_p(t.data_ptr())

    # -- kernels --------------------------------------------------------------------
    def stft(self, x, frames=None):
        """x (B, Lx) -> complex64 (B, T, F).  T defaults to len(range(0, Lx - N, H))."""
        B, Lx = x.shape
        T = self._lib.gl_num_frames(self.n_fft, self.hop, Lx) if frames is None else int(frames)
        if T < 1:
            return torch.empty((B, 0, self.n_bins), dtype=torch.complex64, device=self.device)
        S = torch.empty((B, T, self.n_bins, 2), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_stft(self._h, self._f32(x), Lx, self._f32(S), B, T,
                                         _stream_ptr(self.device)), "gl_stft")
        return torch.view_as_complex(S)

    def istft(self, S, normalize=False):
        """complex64 (B, T, F) -> (B, T*H+N) fp32.  normalize: divide by the window-square sum."""
        self.set_options(window_sum_normalize=normalize)
        if S.dtype != torch.complex64:
            raise ValueError("expected complex64")
        Sr = torch.view_as_real(S.contiguous())
        B, T, F, _ = Sr.shape
        if F != self.n_bins:
            raise ValueError(f"expected {self.n_bins} bins, got {F}")
        y = torch.empty((B, self.signal_length(T)), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_istft(self._h, self._f32(Sr), self._f32(y), B, T,
                                          _stream_ptr(self.device)), "gl_istft")
        return y

    def mel_to_mag(self, filterbank, mel, out=None, clamp=False):
        """filterbank (n_mels, F), mel (B, n_mels, T) -> mag (B, T, F).  clamp: max(0, .)."""
        self.set_options(clamp_mag=clamp)
        B, n_mels, T = mel.shape
        if out is None:
            out = torch.empty((B, T, self.n_bins), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_mel_to_mag(self._h, self._f32(filterbank, (n_mels, self.n_bins)),
                                               self._f32(mel), self._f32(out, (B, T, self.n_bins)), n_mels, B, T,
                                               _stream_ptr(self.device)), "gl_mel_to_mag")
        return out

    def logmel_to_mag(self, filterbank, logmel, compression_factor=1.0, out=None, clamp=False):
        """filterbank (n_mels, F), logmel (B, T, n_mels) natural-log mel (the decoder's layout)
        -> mag (B, T, F); exp, 1/compression_factor and the transpose are fused (infer.py:27-28)."""
        self.set_options(clamp_mag=clamp)
        B, T, n_mels = logmel.shape
        if out is None:
            out = torch.empty((B, T, self.n_bins), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_logmel_to_mag(self._h, self._f32(filterbank, (n_mels, self.n_bins)),
                                                  self._f32(logmel), self._f32(out, (B, T, self.n_bins)), n_mels,
                                                  float(compression_factor), B, T,
                                                  _stream_ptr(self.device)), "gl_logmel_to_mag")
        return out

    def wav_to_logmel(self, y, filterbank, pad_mode="reflect", clip_min=0.01, compression_factor=1.0):
        """y (B, len) -> (log-mel (B, n_mels, T), |stft| (B, T, F)), T = 1 + len // hop (data.py:110-114)."""
        B, n = y.shape
        n_mels = filterbank.shape[0]
        T = int(self._lib.gl_analysis_num_frames(self.hop, n))
        mag = torch.empty((B, T, self.n_bins), dtype=torch.float32, device=self.device)
        logmel = torch.empty((B, n_mels, T), dtype=torch.float32, device=self.device)
        mode = {"constant": 0, "reflect": 1}[pad_mode]
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_wav_to_logmel(self._h, self._f32(y), n, n, self._f32(filterbank, (n_mels, self.n_bins)),
                                                  self._f32(mag), self._f32(logmel), n_mels, mode, float(clip_min),
                                                  float(compression_factor), B, _stream_ptr(self.device)),
                       "gl_wav_to_logmel")
        return logmel, mag

    def resample(self, y, sr_in, sr_out):
        """y (B, n) at sr_in -> (B, ceil(n * sr_out / sr_in)) at sr_out: the conversion librosa.load applies
        (resampy 'kaiser_best', gl_resample)."""
        B, n = y.shape
        n_out = int(self._lib.gl_resample_length(n, int(sr_in), int(sr_out)))
        out = torch.empty((B, n_out), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_resample(self._h, self._f32(y), n, n, self._f32(out), n_out, int(sr_in),
                                             int(sr_out), B, _stream_ptr(self.device)), "gl_resample")
        return out

    def mask_frames(self, mag, lengths):
        """Zero the rows t >= lengths[b] of mag (B, T, F) in place (ragged batches); returns the
        device int32 tensor of lengths."""
        B, T, _ = mag.shape
        lens = torch.as_tensor(lengths, dtype=torch.int32).to(self.device).contiguous()
        if tuple(lens.shape) != (B,):
            raise ValueError(f"lengths must have shape ({B},)")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_mask_frames(self._h, self._f32(mag), ctypes.c_void_p(lens.data_ptr()), B, T,
                                                _stream_ptr(self.device)), "gl_mask_frames")
        return lens

    def run(self, mag, x0, iterations, out=None, normalize=False, momentum=0.0, path=None, run_frames=None):
        """mag (B, T, F), x0 (B, L) -> x (B, L) after `iterations` Griffin-Lim iterations.
        normalize / momentum: window-sum normalised ISTFT, fast Griffin-Lim (not in the reference).
        path / run_frames: kernel family and frames-per-wave-run for THIS call (None = the plan's default);
        they never outlive the call."""
        self.set_options(window_sum_normalize=normalize, momentum=momentum)
        B, T, F = mag.shape
        if F != self.n_bins:
            raise ValueError(f"expected {self.n_bins} bins, got {F}")
        L = self.signal_length(T)
        if out is None:
            out = torch.empty((B, L), dtype=torch.float32, device=self.device)
        with self._scoped(path, run_frames), torch.cuda.device(self.device):
            _lib.check(self._lib.gl_run(self._h, self._f32(mag), self._f32(x0, (B, L)), self._f32(out, (B, L)),
                                        int(iterations), B, T, _stream_ptr(self.device)), "gl_run")
        return out

    def mel_run_supported(self, path=None):
        """True where mel_run exists: the loop of this plan reads packed magnitudes (n_fft 2048, hop
        128/256/512/1024) and the call would take the default path."""
        if (self._path if path is None else int(path)) != 0:
            return False
        return bool(self._lib.gl_mel_run_supported(self._h))

    def mel_run(self, filterbank, mel, x0, iterations, out=None, logmel=False, compression_factor=1.0, clamp=False,
                lengths=None, normalize=False, momentum=0.0, run_frames=None):
        """mel_to_mag / logmel_to_mag (+ mask_frames) + run in one call that never materialises the (B, T, F)
        magnitudes (gl_mel_run): mel (B, n_mels, T) linear, or with logmel=True (B, T, n_mels) natural log.
        lengths: frames per utterance (ragged batch) or None.  Same samples as the separate calls, bit for bit."""
        self.set_options(window_sum_normalize=normalize, clamp_mag=clamp, momentum=momentum)
        if logmel:
            B, T, n_mels = mel.shape
        else:
            B, n_mels, T = mel.shape
        L = self.signal_length(T)
        if out is None:
            out = torch.empty((B, L), dtype=torch.float32, device=self.device)
        lens = None
        if lengths is not None:
            lens = torch.as_tensor(lengths, dtype=torch.int32).to(self.device).contiguous()
            if tuple(lens.shape) != (B,):
                raise ValueError(f"lengths must have shape ({B},)")
        with self._scoped(None, run_frames), torch.cuda.device(self.device):
            _lib.check(self._lib.gl_mel_run(self._h, self._f32(filterbank, (n_mels, self.n_bins)), self._f32(mel),
                                            n_mels, 1 if logmel else 0, float(compression_factor),
                                            ctypes.c_void_p(lens.data_ptr()) if lens is not None else None,
                                            self._f32(x0, (B, L)), self._f32(out, (B, L)), int(iterations), B, T,
                                            _stream_ptr(self.device)), "gl_mel_run")
        return out

    def randn(self, shape, seed, offset=0):
        """Standard normals from the device Philox stream of `seed`, starting at element `offset`."""
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_randn_at(self._f32(out), out.numel(), ctypes.c_uint64(int(seed) & (2**64 - 1)),
                                             int(offset), _stream_ptr(self.device)), "gl_randn_at")
        return out

    def stitch_add(self, out, piece, offset, fade_in=0, fade_out=0):
        """out (L,) += faded piece placed at sample `offset` (gl_stitch_add, long-form seams)."""
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_stitch_add(self._f32(out), out.numel(), self._f32(piece), piece.numel(),
                                               int(offset), int(fade_in), int(fade_out),
                                               _stream_ptr(self.device)), "gl_stitch_add")
        return out

    def peak_normalise(self, x, mode=0, peaks=None):
        """In place on x (B, L); returns the per-row peaks (B,) before scaling (written into
        `peaks` when given -- pass it under capture(), where nothing may be allocated)."""
        B, L = x.shape
        if peaks is None:
            peaks = torch.empty((B,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_peak_normalise(self._h, self._f32(x), L, L, self._f32(peaks), int(mode), B,
                                                   _stream_ptr(self.device)), "gl_peak_normalise")
        return peaks


    def pcm16_encode(self, x):
        """x (B, L) fp32 -> (int16 (B, L) on device, peaks (B,)): the sample conversion of
        save_audio_to_file (audio_utilities.py:202-210), float64 arithmetic, round-half-even."""
        B, L = x.shape
        pcm = torch.empty((B, L), dtype=torch.int16, device=self.device)
        peaks = torch.empty((B,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_pcm16_encode(self._h, self._f32(x), L, L, ctypes.c_void_p(pcm.data_ptr()), L,
                                                 self._f32(peaks), B, _stream_ptr(self.device)), "gl_pcm16_encode")
        return pcm, peaks


    # -- hipGraph capture ---------------------------------------------------------------
    def capture(self):
        """Context manager: the plan calls made inside are recorded into a hipGraph instead of
        running (all tensors, outputs included, must exist beforehand: pass out=...).

            with plan.capture() as g:
                plan.mel_to_mag(fb, mel, out=mag); plan.run(mag, x0, 60, out=x); plan.peak_normalise(x)
            g.launch()          # replays the 64 kernels with one submission on the current stream
        """
        return _Capture(self)


class Graph:
    """An instantiated hipGraph of plan calls (gl_graph_*)."""

    def __init__(self, lib, handle, device):
        self._lib, self._h, self.device = lib, handle, device

    def launch(self):
        with torch.cuda.device(self.device):
            _lib.check(self._lib.gl_graph_launch(self._h, _stream_ptr(self.device)), "gl_graph_launch")

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.gl_graph_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Capture:
    def __init__(self, plan):
        self.plan = plan
        self.graph = Graph(plan._lib, ctypes.c_void_p(), plan.device)

    def __enter__(self):
        p = self.plan
        self._side = torch.cuda.Stream(device=p.device)
        self._side.wait_stream(torch.cuda.current_stream(p.device))
        self._ctx = torch.cuda.stream(self._side)
        self._ctx.__enter__()
        _lib.check(p._lib.gl_graph_begin(ctypes.c_void_p(self._side.cuda_stream)), "gl_graph_begin")
        return self.graph

    def __exit__(self, et, ev, tb):
        p = self.plan
        h = ctypes.c_void_p()
        rc = p._lib.gl_graph_end(ctypes.c_void_p(self._side.cuda_stream), ctypes.byref(h))
        self._ctx.__exit__(et, ev, tb)
        if et is None:
            _lib.check(rc, "gl_graph_end")
            self.graph._h = h
        elif rc == 0:
            p._lib.gl_graph_destroy(h)
        return False


_cache = {}
_cache_lock = threading.Lock()


def get_plan(n_fft, hop, frames, batch, device=None):
    """Cached plan for (n_fft, hop, device), regrown when a call exceeds its bounds.  The scratch is sized from
    the largest frame TOTAL (batch * frames) any call has asked for, not from max(frames) x max(batch): one long
    utterance followed by a wide batch of short ones holds memory for the larger of the two, not their product."""
    dev = _require_gpu(device)
    key = (int(n_fft), int(hop), dev.index)
    frames, batch = int(frames), int(batch)
    with _cache_lock:
        p = _cache.get(key)
        if p is None or p.max_frames < frames or p.max_batch < batch or p.max_total_frames < frames * batch:
            mf = max(frames, p.max_frames if p else 0)
            mb = max(batch, p.max_batch if p else 0)
            cap = max(frames * batch, p.max_total_frames if p else 0)
            if p is not None:
                torch.cuda.synchronize(dev)
                p.close()
            p = Plan(n_fft, hop, mf, mb, dev, max_total_frames=cap)
            p._shared = True
            _cache[key] = p
        return p


def clear_plans():
    with _cache_lock:
        for p in _cache.values():
            p.close()
        _cache.clear()
