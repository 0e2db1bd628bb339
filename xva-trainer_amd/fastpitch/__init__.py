"""FastPitch1.1 on libxvahip: host-side mirror of python/fastpitch1_1/ (model, loss, LAMB, trainer)."""
