"""The boundary as the reference uses it: `import griffin_lim` from dropin/ with the CALLER's top-level
`params` module, then griffin_lim.save_mel_to_wav(mel, "Inference") exactly as infer.py:29 calls it.  Runs in a
subprocess so that sys.path and the module cache are the caller's, not pytest's."""
import os
import subprocess
import sys
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helpers", "infer_call.py")
FIXTURE = os.path.join(ROOT, "tests", "golden", "save_mel_to_wav_cfg1.npz")

# the reference's params.py:36-45,52 (output + audio + misc blocks) with the output directory moved under tmp_path
# and the iteration count of the fixture (the reference default of 200 is a number, not a code path)
PARAMS_PY = """\
use_gpu = True
# output parameters
audio_out_path = {out!r}
mel_out_path = {mel_out!r}
# audio parameters
sampling_rate = 22050
n_mel_channels = 80
frame_hop = 275
f_min = 125
f_max = 7600
n_fft = 1024
griffin_lim_iters = {iters}
# misc parameters
seed = 42
"""


def _caller_dir(tmp_path, iters):
    d = tmp_path / "caller"
    d.mkdir()
    out = tmp_path / "out" / "audio"
    (d / "params.py").write_text(PARAMS_PY.format(out=str(out), mel_out=str(tmp_path / "out" / "mel"), iters=iters))
    return d, out


def _run(caller_dir, mode):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    return subprocess.run([sys.executable, HELPER, str(caller_dir), FIXTURE, mode], capture_output=True, text=True,
                          timeout=600, env=env, cwd=str(caller_dir))


def test_dropin_modules_import_with_the_callers_params(tmp_path):
    """No GPU needed: the shims resolve, the caller's params wins, the module-level filterbank is built from it."""
    caller, _ = _caller_dir(tmp_path, 60)
    r = _run(caller, "import")
    assert r.returncode == 0 and "IMPORT-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_dropin_falls_back_to_packaged_params_when_the_caller_has_none(tmp_path):
    """A `params` module without the audio block (or none at all) must not break the import."""
    d = tmp_path / "caller"
    d.mkdir()
    (d / "params.py").write_text("use_gpu = False\n")
    code = ("import sys; sys.path[:0] = [%r, %r]; import griffin_lim, tacotron2_amd.params as tp; "
            "assert griffin_lim.params is tp; print('FALLBACK-OK')" % (os.path.join(ROOT, "dropin"), str(d)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(d))
    assert r.returncode == 0 and "FALLBACK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_infer_py_call_path_writes_the_reference_wav(tmp_path, golden):
    d = golden("save_mel_to_wav_cfg1")
    caller, out_dir = _caller_dir(tmp_path, int(d["iters"]))
    r = _run(caller, "run")
    assert r.returncode == 0 and "RUN-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    # save_audio_to_file prints max and min of the samples first (audio_utilities.py:200-201)
    lines = r.stdout.split()
    assert float(lines[0]) > 0.0 > float(lines[1])
    wav = out_dir / "Inference.wav"
    assert wav.exists(), "Inference.wav must appear under the caller's params.audio_out_path"
    w = wave.open(str(wav))
    assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getcomptype()) == (1, 2, 22050, "NONE")
    pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    assert pcm.shape == d["pcm"].shape
    assert np.max(np.abs(pcm.astype(int) - d["pcm"].astype(int))) <= 34        # 1e-3 of full scale (north_star)
    ret = np.load(str(caller / "returned.npy"))
    assert ret.shape == (513, d["mel"].shape[1])
