import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import fastpitch as ofp
from xva_trainer_amd import _lib
from xva_trainer_amd.fastpitch import params as P
from xva_trainer_amd.fastpitch.engine import DeviceBatch
from fp_util import build_engine
sd = ofp.init_state_dict(13); batch = ofp.synth_batch(3, 41, 300, 6)
res = {}
for name, prod, planes in (("exact", 0, 1), ("planes", 1, 1)):
    _lib.lib.xva_gemm_set_fp32_products(prod); _lib.lib.xva_fp_set_ffn_planes(planes)
    eng, flat, grads = build_engine(sd, "fp32")
    b = DeviceBatch.from_dict(batch, "cuda")
    grads.zero_(); eng.fwd_loss_bwd(flat, grads, b, 3); torch.cuda.synchronize()
    res[name] = {k: v.clone() for k, v in P.from_flat(grads, eng.table).items()}
for a in ("planes",):
    grp = collections.defaultdict(list)
    for k in res["exact"]:
        ref = res["exact"][k].double()
        if float(ref.abs().max()) == 0: continue
        key = ".".join(k.split(".")[:2]) if k.startswith(("encoder.layers", "decoder.layers")) else k.split(".")[0]
        grp[key].append(float((res[a][k].double() - ref).norm() / ref.norm()))
    print("==", a, "vs exact:", {k: "%.1e" % max(v) for k, v in sorted(grp.items())})
print("---- per tensor planes vs exact, table order")
for name, off, n, shape, kind in eng.table:
    if name not in res["exact"]: continue
    ref = res["exact"][name].double()
    if float(ref.abs().max()) == 0: continue
    r = float((res["planes"][name].double() - ref).norm() / ref.norm())
    if name.startswith(("pitch_", "energy_", "duration_")) or "layers.5" in name and name.startswith("encoder"): print("%-52s off %9d n %8d rel %.2e" % (name, off, n, r))
k = "energy_predictor.layers.1.conv.bias"
e, pl_ = res["exact"][k].double().cpu(), res["planes"][k].double().cpu()
d = (pl_ - e)
print("c2_b: exact norm %.4e diff norm %.4e; top diffs idx %s vals %s ; exact there %s" % (e.norm(), d.norm(), d.abs().topk(6).indices.tolist(), [round(float(v), 6) for v in d[d.abs().topk(6).indices]], [round(float(v), 6) for v in e[d.abs().topk(6).indices]]))
print("ratio stats: mean %.6f std %.6f" % (float((pl_ / e).mean()), float((pl_ / e).std())))
