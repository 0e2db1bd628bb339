"""xVAPitch-only blocks on libxvahip: WaveNet gated stack, residual coupling blocks, posterior encoder (wn.py), relative-position
transformer (transformer.py), stochastic duration predictor (sdp.py), monotonic alignment search / segment gather / KL loss (ops.py), and the
acoustic half of the generator train step that composes them (acoustic.py).  The shared blocks (HiFi-GAN generator / discriminators, mel front
ends) live in xva-trainer_amd/hifigan and xva-trainer_amd/mel.py."""
