"""Top-level ``griffin_lim`` module for the reference's ``infer.py``
(``import griffin_lim``, infer.py:5): re-exports the MI355X implementation.
Put this directory (and the repo root) ahead of the reference on sys.path."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(1, _root)

from tacotron2_amd.griffin_lim import filterbank, mel_to_audio, params, save_mel_to_wav  # noqa: E402,F401
