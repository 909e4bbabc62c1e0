"""ctypes binding of libgl_hip.so (C-ABI declared in include/gl_hip.h).

The library is built in-tree by ``tacotron2_amd/csrc/Makefile`` (or
``__graft_entry__.build()``).  There is no CPU fallback: if the shared object is
missing or a symbol is absent, importing this module's ``lib()`` raises.

The one environment variable the package reads is ``GL_HIP_LIB``: the path of another build of the
same library (the kernel-study builds of ``abl/build.sh``).  The HIP sources read none.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GL_HIP_LIB") or os.path.join(_HERE, "libgl_hip.so")   # override: kernel experiments

_c_float_p = ctypes.c_void_p   # device pointers travel as integers
_plan_p = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/gl_hip.h one to one
SIGNATURES = {
    "gl_version": (ctypes.c_int, []),
    "gl_last_error": (ctypes.c_char_p, []),
    "gl_signal_length": (ctypes.c_longlong, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "gl_num_frames": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_longlong]),
    "gl_plan_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(_plan_p)]),
    "gl_plan_create_capacity": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(_plan_p)]),
    "gl_plan_destroy": (ctypes.c_int, [_plan_p]),
    "gl_plan_device_bytes": (ctypes.c_longlong, [_plan_p]),
    "gl_plan_set_path": (ctypes.c_int, [_plan_p, ctypes.c_int]),
    "gl_plan_set_run_frames": (ctypes.c_int, [_plan_p, ctypes.c_int]),
    "gl_plan_set_options": (ctypes.c_int, [_plan_p, ctypes.c_int, ctypes.c_float]),
    "gl_stft": (ctypes.c_int, [_plan_p, _c_float_p, ctypes.c_longlong, _c_float_p, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p]),
    "gl_istft": (ctypes.c_int, [_plan_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "gl_mel_to_mag": (ctypes.c_int, [_plan_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p]),
    "gl_logmel_to_mag": (ctypes.c_int, [_plan_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "gl_analysis_num_frames": (ctypes.c_int, [ctypes.c_int, ctypes.c_longlong]),
    "gl_wav_to_logmel": (ctypes.c_int, [_plan_p, _c_float_p, ctypes.c_longlong, ctypes.c_longlong, _c_float_p,
                                        _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_int, ctypes.c_void_p]),
    "gl_run": (ctypes.c_int, [_plan_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_void_p]),
    "gl_mel_run_supported": (ctypes.c_int, [_plan_p]),
    "gl_mel_run": (ctypes.c_int, [_plan_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                  ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p]),
    "gl_resample_length": (ctypes.c_longlong, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int]),
    "gl_resample": (ctypes.c_int, [_plan_p, _c_float_p, ctypes.c_longlong, ctypes.c_longlong, _c_float_p,
                                   ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "gl_graph_begin": (ctypes.c_int, [ctypes.c_void_p]),
    "gl_graph_end": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "gl_graph_launch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "gl_graph_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "gl_mask_frames": (ctypes.c_int, [_plan_p, _c_float_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "gl_randn": (ctypes.c_int, [_c_float_p, ctypes.c_longlong, ctypes.c_uint64, ctypes.c_void_p]),
    "gl_randn_at": (ctypes.c_int, [_c_float_p, ctypes.c_longlong, ctypes.c_uint64, ctypes.c_longlong, ctypes.c_void_p]),
    "gl_stitch_add": (ctypes.c_int, [_c_float_p, ctypes.c_longlong, _c_float_p, ctypes.c_longlong, ctypes.c_longlong,
                                     ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p]),
    "gl_peak_normalise": (ctypes.c_int, [_plan_p, _c_float_p, ctypes.c_longlong, ctypes.c_longlong, _c_float_p,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "gl_peak_normalise_static": (ctypes.c_int, [_c_float_p, ctypes.c_longlong, ctypes.c_longlong, _c_float_p,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "gl_pcm16_encode": (ctypes.c_int, [_plan_p, _c_float_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p,
                                       ctypes.c_longlong, _c_float_p, ctypes.c_int, ctypes.c_void_p]),
}

_lib = None


class GLError(RuntimeError):
    """A libgl_hip call returned a negative status."""


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C tacotron2_amd/csrc). "
            "There is no CPU fallback for the Griffin-Lim path.")
    # PyTorch-ROCm ships its own libamdhip64; load it first so that this library binds
    # to the SAME HIP runtime (streams and device pointers are shared with torch).
    import torch  # noqa: F401
    handle = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def check(status, what):
    """Raise on a negative gl_status: ValueError for argument errors, GLError otherwise."""
    if status >= 0:
        return status
    msg = lib().gl_last_error().decode("utf-8", "replace")
    text = f"{what} failed (status {status}): {msg}"
    if status == -1:
        raise ValueError(text)
    if status == -3:
        raise MemoryError(text)
    raise GLError(text)
