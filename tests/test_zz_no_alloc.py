"""gl_peak_normalise_static (the entry point that allocates nothing) against gl_peak_normalise, through
the C-ABI: same kernels, so outputs and peaks must be bit-identical.  Two calls in a row, the second
on a smaller input (the static array keeps the first call's maxima).  Runs last (file name)."""
import ctypes

import numpy as np
import pytest
import torch

from tacotron2_amd import _lib
from tacotron2_amd.plan import Plan

pytestmark = pytest.mark.gpu


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _both(lib, plan, x, mode):
    """(existing entry point's result, the static one's) on clones of x: (x_scaled, peaks) each."""
    B, L = x.shape
    dev = x.device
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    xa, xb = x.clone(), x.clone()
    pa = torch.full((B,), -1.0, dtype=torch.float32, device=dev)
    pb = torch.full((B,), -2.0, dtype=torch.float32, device=dev)
    _lib.check(lib.gl_peak_normalise(plan._h, _ptr(xa), L, L, _ptr(pa), mode, B, stream), "gl_peak_normalise")
    _lib.check(lib.gl_peak_normalise_static(_ptr(xb), L, L, _ptr(pb), mode, B, dev.index, stream),
               "gl_peak_normalise_static")
    torch.cuda.synchronize()
    return (xa, pa), (xb, pb)


def test_static_peak_normalise_matches_plan_entry_point_twice():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    plan = Plan(64, 16, 4, 64, dev)
    rng = np.random.default_rng(11)
    # first call: 37 rows, peaks on both sides of 1 (mode 0 scales only rows whose peak exceeds 1), one all-zero row
    big = rng.standard_normal((37, 50021)).astype(np.float32) * rng.uniform(0.05, 3.0, size=(37, 1)).astype(np.float32)
    big[5] = 0.0
    big[9, 123] = 7.5
    # second call: fewer, shorter rows with SMALLER peaks than the first call left in the array
    small = (rng.standard_normal((3, 4097)) * 0.01).astype(np.float32)
    small[1, 0] = -1.25
    for mode in (0, 1):
        for host in (big, small):
            x = torch.from_numpy(host).to(dev)
            (xa, pa), (xb, pb) = _both(lib, plan, x, mode)
            assert torch.equal(pa, pb), "peaks differ"
            assert torch.equal(xa, xb), "scaled samples differ"
            ref_peak = np.max(np.abs(host), axis=1)
            assert np.array_equal(pa.cpu().numpy(), ref_peak)
            # the reference's rule (griffin_lim.py:27-29): divide by the peak only when it exceeds 1 (mode 0)
            scale = np.where((ref_peak > 1.0) if mode == 0 else (ref_peak > 0.0), ref_peak, 1.0).astype(np.float32)
            assert np.array_equal(xb.cpu().numpy(), host / scale[:, None])
    # a strided view of rows (row_stride > len) and no peaks output
    x = torch.from_numpy(big).to(dev)
    L = 1000
    xa, xb = x.clone(), x.clone()
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.gl_peak_normalise(plan._h, _ptr(xa), big.shape[1], L, None, 1, 4, stream), "gl_peak_normalise")
    _lib.check(lib.gl_peak_normalise_static(_ptr(xb), big.shape[1], L, None, 1, 4, 0, stream), "static")
    torch.cuda.synchronize()
    assert torch.equal(xa, xb)
    plan.close()


def test_static_peak_normalise_rejects_bad_arguments():
    lib = _lib.lib()
    x = torch.zeros((2, 64), dtype=torch.float32, device="cuda:0")
    limit = 4096       # GL_STATIC_MAX_BATCH
    assert lib.gl_peak_normalise_static(_ptr(x), 64, 64, None, 0, limit + 1, 0, None) == -1
    assert b"GL_STATIC_MAX_BATCH" in lib.gl_last_error()
    assert lib.gl_peak_normalise_static(None, 64, 64, None, 0, 2, 0, None) == -1
    assert lib.gl_peak_normalise_static(_ptr(x), 32, 64, None, 0, 2, 0, None) == -1      # stride < len
    assert lib.gl_peak_normalise_static(_ptr(x), 64, 64, None, 2, 2, 0, None) == -1      # bad mode
    assert lib.gl_peak_normalise_static(_ptr(x), 64, 64, None, 0, 2, 99, None) < 0       # no such device


def test_static_peak_normalise_inside_stream_capture():
    """The point of the entry point: legal under capture with no plan and nothing prepared."""
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    host = (rng.standard_normal((6, 30000)) * 2.0).astype(np.float32)
    x = torch.from_numpy(host).to(dev)
    peaks = torch.zeros((6,), dtype=torch.float32, device=dev)
    # first use on the device outside capture (symbol lookup), like a library init
    warm = x.clone()
    _lib.check(lib.gl_peak_normalise_static(_ptr(warm), 30000, 30000, None, 0, 6, 0,
                                            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "warm")
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    sp = ctypes.c_void_p(side.cuda_stream)
    g = ctypes.c_void_p()
    _lib.check(lib.gl_graph_begin(sp), "gl_graph_begin")
    rc = lib.gl_peak_normalise_static(_ptr(x), 30000, 30000, _ptr(peaks), 0, 6, 0, sp)
    rc_end = lib.gl_graph_end(sp, ctypes.byref(g))
    _lib.check(rc, "gl_peak_normalise_static under capture")
    _lib.check(rc_end, "gl_graph_end")
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.from_numpy(host)), "capture must not run the kernels"
    _lib.check(lib.gl_graph_launch(g, sp), "gl_graph_launch")
    torch.cuda.synchronize()
    assert torch.equal(x, warm)
    assert np.array_equal(peaks.cpu().numpy(), np.max(np.abs(host), axis=1))
    lib.gl_graph_destroy(g)
