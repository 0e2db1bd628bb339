import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import fastpitch as ofp
from xva_trainer_amd import _lib
from xva_trainer_amd.fastpitch.engine import DeviceBatch
from fp_util import build_engine
sd = ofp.init_state_dict(13); batch = ofp.synth_batch(3, 41, 300, 6)
_lib.lib.xva_gemm_set_fp32_products(1)
eng, flat, grads = build_engine(sd, "fp32")
b = DeviceBatch.from_dict(batch, "cuda")
for it in range(2):
    eng.forward(flat, b, 3); torch.cuda.synchronize()
    for stack, T in (("encoder", 41), ("decoder", int(batch["mel_lens"].max()))):
        for l in range(7):
            x = eng.layer_input(stack, l, 3, T)
            n = torch.isnan(x).any(dim=2)
            print(it, stack, l, "nan rows per item:", n.sum(1).tolist(), "first", [int(r.nonzero()[0]) if r.any() else -1 for r in n])
print("---- with backward")
for it in range(2):
    grads.zero_(); eng.fwd_loss_bwd(flat, grads, b, 3); torch.cuda.synchronize()
    print(it, "grads nan:", int(torch.isnan(grads).sum()), "mel nan:", int(torch.isnan(eng.outputs(b, 3)["mel_out"]).sum()))
    for stack, T in (("encoder", 41), ("decoder", int(batch["mel_lens"].max()))):
        for l in range(7):
            x = eng.layer_input(stack, l, 3, T)
            n = torch.isnan(x).any(dim=2)
            if n.any(): print(it, stack, l, "nan rows per item:", n.sum(1).tolist())
