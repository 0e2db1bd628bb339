#!/bin/bash
# Round-end evidence on the GPU box: plain bench line, rocprofv3 kernel stats of the same command, and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs as MI355X_MICROARCH.md prescribes) per leg.  Everything lands in gpurun_out/final/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final; mkdir -p $O
cd $R && python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_under_rocprof.json
python $R/tools/rocpd_summary.py $(find /tmp/p_stats -name "*.db" | head -1) $O/bench_kernel_stats.csv
for leg in fp hg; do
  if [ $leg = fp ]; then CMD="python $R/bench.py --steps 2 --warmup 1 --no-hifigan --no-cpu-baseline --no-roofline"; else CMD="python $R/tools/hg_gemm_profile.py 64 noprof"; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -d /tmp/p_${leg}_$ctr -o c -- $CMD > /dev/null 2>&1
    python $R/tools/pmc_summary.py $(find /tmp/p_${leg}_$ctr -name "*.db" | head -1) /tmp/${leg}_$ctr.csv
  done
  python $R/tools/pmc_merge.py /tmp/${leg}_FETCH_SIZE.csv /tmp/${leg}_WRITE_SIZE.csv $O/${leg}_pmc_hbm_bytes.csv
done
ls -la $O
