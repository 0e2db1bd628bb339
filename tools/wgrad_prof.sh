#!/bin/bash
# per-shape kernel durations of tools/wgrad_bench.py under rocprofv3 (the bench's own event timing has a ~10 us host floor per call)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for s in ${1:-gen_c32_k11 gen_c32_k3 gen_c64_k11 gen_c64_k7 gen_c128_k11 gen_c128_k3 msd_conv1 msd_conv2 msd_conv3 msd_conv4 msd_conv5}; do
  rm -rf /tmp/p_$s
  rocprofv3 --kernel-trace --stats -d /tmp/p_$s -o x -- python $R/tools/wgrad_bench.py $s $2 > /tmp/o_$s.txt 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/p_$s -name "*.db" | head -1) /tmp/s_$s.csv > /dev/null
  echo "== $s"; grep -E "wgrad|reduce_kernel|xva_gemm_glds" /tmp/s_$s.csv | awk -F'"' '{n=split($0,a,","); print $2, a[n-3], a[n-1]}' | sed 's/(xva_gemm_params.*)//'
  grep -E "wgrad|reduce_kernel|xva_gemm_glds" /tmp/s_$s.csv | grep -v '^"' | awk -F, '{print $1,$2,$4}'
done
