"""Top-level ``audio_utilities`` module (reference griffin_lim.py:6 does
``import audio_utilities``): re-exports the MI355X implementation."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(1, _root)

from tacotron2_amd.audio_utilities import *  # noqa: E402,F401,F403
from tacotron2_amd.audio_utilities import __all__  # noqa: E402,F401
