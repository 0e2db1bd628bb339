"""Per-shape timing of every xva_gemm launch in one FastPitch fwd+bwd at the bench configuration (HIP events around each launch)."""
import sys, os, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib, synthetic
from xva_trainer_amd.fastpitch import engine as E, params as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MODE = os.environ.get("XVA_FP_MODE", "bf16")     # bf16 | f16 (fp16 operands, fp32 residual stream) | split (fp32 storage, split-bf16 products on pairs)
if MODE == "split": _lib.lib.xva_gemm_set_fp32_products(1)
eng = E.FastPitchEngine("cuda", {"split": "fp32", "f16": "f16"}.get(MODE, "bf16"), p_dropout=0.1)
flat = torch.zeros(eng.total, device="cuda"); P.default_init_(flat, eng.table, seed=1234)
grads = torch.zeros_like(flat)
batch = E.DeviceBatch.from_dict(synthetic.fastpitch_batch(B, 150, 860, 1234), "cuda")
for _ in range(2):
    eng.fwd_loss_bwd(flat, grads, batch, 3)
torch.cuda.synchronize()
_lib.lib.xva_prof_enable(1)
eng.fwd_loss_bwd(flat, grads, batch, 3); torch.cuda.synchronize()
_lib.lib.xva_prof_enable(0)
os.makedirs("gpurun_out", exist_ok=True)
_lib.lib.xva_prof_dump(b"gpurun_out/fp_gemm_launches.csv")
rows = list(csv.DictReader(open("gpurun_out/fp_gemm_launches.csv")))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    k = (r["variant"], r["M"], r["N"], r["K"], r["batch"], r["splitk"], r["bn"])
    agg[k][0] += 1; agg[k][1] += float(r["ms"]); agg[k][2] += float(r["gflop"])
tot = sum(v[1] for v in agg.values())
print("total GEMM ms", tot, "launches", len(rows))
names = ["NT", "NN", "TN"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('XVA_TOP', '40'))]:
    print("%s m%d M=%-6s N=%-5s K=%-7s batch=%-5s sk=%-3s bn=%-3s n=%-3d ms=%8.3f  TF=%7.1f" % (names[int(k[0]) // 3], int(k[0]) % 3, k[1], k[2], k[3], k[4], k[5], k[6], v[0], v[1], v[2] / v[1] if v[1] else 0))
