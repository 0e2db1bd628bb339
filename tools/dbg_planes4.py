import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import fastpitch as ofp
from xva_trainer_amd import _lib
from xva_trainer_amd.fastpitch.engine import DeviceBatch
from fp_util import build_engine
sd = ofp.init_state_dict(13); batch = ofp.synth_batch(3, 41, 300, 6)
res = {}
for name, prod in (("exact", 0), ("planes", 1)):
    _lib.lib.xva_gemm_set_fp32_products(prod)
    eng, flat, grads = build_engine(sd, "fp32")
    b = DeviceBatch.from_dict(batch, "cuda")
    grads.zero_(); losses = eng.fwd_loss_bwd(flat, grads, b, 3); torch.cuda.synchronize()
    o = eng.outputs(b, 3)
    res[name] = ({k: v.float().clone() for k, v in o.items() if torch.is_tensor(v)}, losses.cpu().clone())
for k in res["exact"][0]:
    a, r = res["planes"][0][k].double(), res["exact"][0][k].double()
    print("%-14s %.2e" % (k, float((a - r).abs().max() / r.abs().max().clamp_min(1e-30))))
print(res["exact"][1].tolist()); print(res["planes"][1].tolist())
