#!/usr/bin/env python3
"""TEST / BASELINE INFRASTRUCTURE -- not part of the product path.

One worker of bench.py's all-cores CPU baseline leg (CPU baseline (ii)): runs the float64
loop-form restatement of audio_utilities.py:154-188 (gl_oracle.griffin_lim_loop, incl. the dead RMSE)
on `reps` utterances of the benchmark geometry, one after the other, single-threaded, and prints the
seconds it took.  bench.py starts one of these per host core.

    python oracle/cpu_worker.py N H T iters reps seed
"""
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ[_v] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402
import gl_oracle as orc  # noqa: E402


def main():
    N, H, T, iters, reps, seed = (int(a) for a in sys.argv[1:7])
    rng = np.random.default_rng(seed)
    mag = np.abs(rng.standard_normal((T, N // 2 + 1))).astype(np.float32)
    x0 = rng.standard_normal(T * H + N)
    orc.griffin_lim_loop(mag, N, H, 1, x0=x0)          # page in numpy/pocketfft before the clock starts
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.griffin_lim_loop(mag, N, H, iters, x0=x0)
    print(f"{time.perf_counter() - t0:.6f}")


if __name__ == "__main__":
    main()
