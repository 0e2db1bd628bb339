"""xVAPitch-only blocks on libxvahip (first set): WaveNet gated stack, residual coupling block, monotonic alignment
search, segment gather, KL loss.  See ops.py / wn.py; the shared blocks (HiFi-GAN generator / discriminators, mel front ends) live in
xva-trainer_amd/hifigan and xva-trainer_amd/mel.py."""
