"""Subprocess body of tests/test_dropin_boundary.py: the reference's own call path, literally.

infer.py:1,5 does `import params` and `import griffin_lim` from the top of sys.path, seeds numpy's global RNG
through set_seed (infer.py:34 -> data.py:22-26) and calls griffin_lim.save_mel_to_wav(mel, "Inference") with the
transposed, non-contiguous (80, T) view of infer.py:28.  sys.path here is [dropin/, <dir holding the caller's
params.py>, ...]: nothing from tacotron2_amd is imported by name.

argv: <dir with params.py> <fixture .npz> <mode: import|run>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
params_dir, fixture, mode = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path[:0] = [os.path.join(ROOT, "dropin"), params_dir]

import numpy as np  # noqa: E402

import params  # noqa: E402      (infer.py:1 -- the caller's own configuration module)
import griffin_lim  # noqa: E402 (infer.py:5)
import audio_utilities  # noqa: E402 (griffin_lim.py:6)

assert os.path.dirname(os.path.abspath(params.__file__)) == os.path.abspath(params_dir)
assert griffin_lim.params is params, "the caller's top-level params module must win"
assert os.path.dirname(os.path.abspath(griffin_lim.__file__)) == os.path.join(ROOT, "dropin")
assert os.path.dirname(os.path.abspath(audio_utilities.__file__)) == os.path.join(ROOT, "dropin")
# the reference's public names (griffin_lim.py:13,17; audio_utilities.py:14-215)
assert griffin_lim.filterbank.shape == (params.n_mel_channels, params.n_fft // 2 + 1)
assert griffin_lim.filterbank.dtype == np.float32
for name in ("hz_to_mel", "mel_to_hz", "fft_bin_to_hz", "hz_to_fft_bin", "stft_for_reconstruction",
             "istft_for_reconstruction", "get_signal", "reconstruct_signal_griffin_lim", "save_audio_to_file"):
    assert callable(getattr(audio_utilities, name)), name
if mode == "import":
    print("IMPORT-OK")
    sys.exit(0)

d = np.load(fixture, allow_pickle=False)
mels = np.ascontiguousarray(d["mel"].T)[None]          # (1, T, 80): what dynamic_range_decompression returns
mel = mels[0].T                                        # infer.py:28 -- a transposed view
assert mel.shape == (80, mels.shape[1]) and not mel.flags["C_CONTIGUOUS"]
np.random.seed(int(d["seed"]))                         # infer.py:34 -> data.py:24
before = mel.copy()
ret = griffin_lim.save_mel_to_wav(mel, "Inference")    # infer.py:29
assert np.array_equal(mel, before), "the caller's array was modified"
assert isinstance(ret, np.ndarray) and ret.shape == (params.n_fft // 2 + 1, mel.shape[1]), ret.shape
lin = np.dot(griffin_lim.filterbank.T.astype(np.float64), mel.astype(np.float64))      # griffin_lim.py:18
assert np.max(np.abs(ret - lin)) <= 1e-5 * np.max(lin)
np.save(os.path.join(params_dir, "returned.npy"), ret)
print("RUN-OK")
