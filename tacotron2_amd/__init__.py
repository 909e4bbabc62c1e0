"""tacotron2_amd: MI355X-native Griffin-Lim vocoder, a drop-in for the
griffin_lim.py / audio_utilities.py path of AidenMoosa/Tacotron2.

    from tacotron2_amd import griffin_lim, audio_utilities

or put ``dropin/`` on sys.path so the reference's ``infer.py`` resolves
``import griffin_lim`` to this implementation unchanged.
"""
__version__ = "0.1.0"

__all__ = ["audio_utilities", "griffin_lim", "params", "filters", "plan"]
