"""Phase breakdown of one HiFi-GAN D+G iteration at BASELINE configs[2] (B = 64 x 8192): HIP-event time of each engine call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
from xva_trainer_amd.hifigan.step import HifiganStep
from xva_trainer_amd import mel as pmel
st = HifiganStep("cuda:0", "bf16")
a = type("A", (), {"compute": "bf16", "hg_batch": B, "hg_steps": 1, "steps": 1, "no_roofline": True})()
bench.init_hifigan_weights(st)
x, y, y_mel = bench.hifigan_inputs(B, 0, "cuda:0")
eng = st.eng
names, evs = [], []
def mark(n):
    e = torch.cuda.Event(enable_timing=True); e.record(); names.append(n); evs.append(e)
def step():
    names.clear(); evs.clear()
    mark("start")
    yg = eng.generator_forward(st.flat_g, x); mark("gen_fwd")
    ld = eng.disc_forward(st.flat_d, y, yg); mark("disc_fwd_1")
    st.grads_d.zero_(); eng.disc_backward_d(st.flat_d, st.grads_d); mark("disc_bwd_d")
    st.optim_d.step(st.grads_d); mark("adamw_d")
    lg = eng.disc_forward(st.flat_d, y, yg); mark("disc_fwd_2")
    dw = eng.disc_backward_g(st.flat_d); mark("disc_bwd_g")
    pmel.mel_l1_loss_backward(yg, y_mel, dw, scale=45.0, accumulate=True); mark("mel_l1")
    st.grads_g.zero_(); eng.generator_backward(st.flat_g, st.grads_g, dw); mark("gen_bwd")
    st.optim_g.step(st.grads_g); mark("adamw_g")
for _ in range(3): step()
acc = {}
N = 5
for _ in range(N):
    step(); torch.cuda.synchronize()
    for i in range(1, len(evs)):
        acc[names[i]] = acc.get(names[i], 0.0) + evs[i - 1].elapsed_time(evs[i])
tot = sum(acc.values())
for k, v in acc.items(): print("%-12s %7.3f ms  %5.1f %%" % (k, v / N, 100 * v / tot))
print("total        %7.3f ms" % (tot / N))
