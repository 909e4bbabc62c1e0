"""Audio / output configuration read by the Griffin-Lim path.

Same names and values as the audio block of the reference's params.py:36-45,52
(the only names griffin_lim.py:7,13-14,23-24,32 reads).  When the reference's own
``params`` module is importable first on sys.path (e.g. running its infer.py
unchanged), ``tacotron2_amd.griffin_lim`` uses that module instead of this one.
"""
# output parameters                       (reference params.py:36-37)
audio_out_path = 'out/audio'
mel_out_path = 'out/mel'
# audio parameters                        (reference params.py:39-45)
sampling_rate = 22050
n_mel_channels = 80
frame_hop = 275
f_min = 125
f_max = 7600
n_fft = 1024
griffin_lim_iters = 200
# misc parameters                         (reference params.py:52)
seed = 42
