cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1; do
  XVA_FP_BWD_NT=$m XVA_FP_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/p_nt$m -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hifigan --no-xvapitch --no-fp32-parity --no-roofline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/p_nt$m -name "*.db" | head -1) $R/gpurun_out/nt_ab_$m.csv > /dev/null
  echo "== NT=$m"; head -12 $R/gpurun_out/nt_ab_$m.csv | cut -c1-150
done
for m in 0 1 0 1; do XVA_FP_BWD_NT=$m python $R/bench.py --steps 40 --warmup 5 --no-hifigan --no-xvapitch --no-fp32-parity --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('NT=$m', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; done
