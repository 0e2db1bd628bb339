#!/bin/bash
# The fused ResBlock pair (conv_pair.hip) against the two launches: time of the generator forward, per-kernel times of its convolutions (rocprofv3 kernel
# trace) and the HBM traffic of the same calls (FETCH_SIZE / WRITE_SIZE, separate --pmc passes).  Output: gpurun_out/pair_ab.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
for m in 0 1 2 3 0 1 2 3; do python $R/tools/pair_bench.py $m 2>/dev/null | tail -1; done
for m in 0 1 3; do
  echo "== per kernel, XVA_HG_PAIR=$m (rocprofv3 --kernel-trace --stats; 13 forwards: Calls / 13 per forward)"
  rm -rf /tmp/pp_$m; rocprofv3 --kernel-trace --stats -d /tmp/pp_$m -o p -- python $R/tools/pair_bench.py $m 10 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pp_$m -name "*.db" | head -1) /tmp/pp_$m.csv > /dev/null
  grep -E "conv_res_kernel<0, (32|64),|conv_pair" /tmp/pp_$m.csv
  python - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/pp_$m.csv")))
sel = [r for r in rows if "conv_pair" in r["Name"] or "conv_res_kernel<0, 32," in r["Name"] or "conv_res_kernel<0, 64," in r["Name"]]
print("   32 / 64-channel forward convolutions: %.1f us per forward; all kernels %.1f us per forward" % (sum(float(r["TotalDurationUs"]) for r in sel) / 13, sum(float(r["TotalDurationUs"]) for r in rows) / 13))
PY
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pc_${m}_$ctr; rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pc_${m}_$ctr -o c -- python $R/tools/pair_bench.py $m 2 > /dev/null 2>&1
    python $R/tools/pmc_summary.py $(find /tmp/pc_${m}_$ctr -name "*.db" | head -1) /tmp/pc_${m}_$ctr.csv > /dev/null
  done
  python - <<PY
import csv
tot = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("/tmp/pc_${m}_%s.csv" % ctr)))
    sel = [r for r in rows if "conv_pair" in r["Kernel"] or "conv_res_kernel<0, 32," in r["Kernel"] or "conv_res_kernel<0, 64," in r["Kernel"]]
    # FETCH_SIZE / WRITE_SIZE count KB on gfx950 as the guide's HBM section says (FETCH_SIZE x 2: 64-byte units reported as 32): tools/pmc_merge.py applies the same
    f = 2.0 if ctr == "FETCH_SIZE" else 1.0
    tot[ctr] = sum(float(r["Total"]) for r in sel) * f / 1000.0 / 5      # 3 warm-up + 2 timed forwards
print("   HBM traffic of the 32 / 64-channel forward convolutions per forward: fetch %.0f MB + write %.0f MB = %.0f MB" % (tot["FETCH_SIZE"], tot["WRITE_SIZE"], tot["FETCH_SIZE"] + tot["WRITE_SIZE"]))
PY
done
} > $O/pair_ab.txt 2>&1
