"""HiFi-GAN v1 on libxvahip: host-side mirror of python/hifigan/ (models, losses, trainer)."""
