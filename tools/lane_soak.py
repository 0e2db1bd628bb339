import sys, os, numpy as np, torch
sys.path.insert(0, ".")
from tests.test_stream_lanes_gpu import _hifigan_step, _fastpitch_step
gd = "tests/golden"
ref = _hifigan_step(gd, 1, "fp32")
worst = 0.0
for it in range(12):
    b = _hifigan_step(gd, 3, "fp32")
    assert torch.equal(ref[0]["y_g_hat"], b[0]["y_g_hat"])
    r = max(float((ref[i] - b[i]).norm() / ref[i].norm()) for i in (1, 2))
    worst = max(worst, r)
print("hifigan fp32 lanes x12: worst rel grad diff", worst)
l0, g0, o0 = _fastpitch_step(1, "fp32")
worst = 0.0
for it in range(12):
    l1, g1, o1 = _fastpitch_step(3, "fp32")
    for k in o0: assert torch.equal(o0[k], o1[k]), k
    r = max(float((g0[k] - g1[k]).norm() / g0[k].norm().clamp_min(1e-30)) for k in g0 if g0[k].abs().max() > 0)
    worst = max(worst, r)
print("fastpitch fp32 lanes x12: worst rel grad diff", worst)
