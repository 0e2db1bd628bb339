import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import fastpitch as ofp
from xva_trainer_amd import _lib
from xva_trainer_amd.fastpitch import params as P
from xva_trainer_amd.fastpitch.engine import DeviceBatch
from fp_util import build_engine
sd = ofp.init_state_dict(13); batch = ofp.synth_batch(3, 41, 300, 6)
_lib.lib.xva_gemm_set_fp32_products(1)
res = {}
MODES = [int(v) for v in os.environ.get('MODES', '1,0').split(',')]
for idx, mode in enumerate(MODES):
    _lib.lib.xva_fp_set_ffn_planes(mode)
    eng, flat, grads = build_engine(sd, "fp32")
    b = DeviceBatch.from_dict(batch, "cuda")
    grads.zero_(); eng.fwd_loss_bwd(flat, grads, b, 3); torch.cuda.synchronize()
    res[1 - idx] = {k: v.clone() for k, v in P.from_flat(grads, eng.table).items()}
w = sorted(((float((res[1][k].double() - res[0][k].double()).norm() / res[0][k].double().norm().clamp_min(1e-30)), k) for k in res[0] if float(res[0][k].abs().max()) > 0), reverse=True)
for r, k in w[:25]: print("%.3e %s" % (r, k))
import collections
grp = collections.defaultdict(list)
for r, k in w:
    key = ".".join(k.split(".")[:3]) if k.startswith(("encoder.layers", "decoder.layers")) else k.split(".")[0]
    grp[key].append(r)
for key in sorted(grp): print("%-28s max %.2e median %.2e n %d" % (key, max(grp[key]), sorted(grp[key])[len(grp[key]) // 2], len(grp[key])))
