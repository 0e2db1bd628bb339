R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_fp_fused_gpu.py -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
XVA_FP_ONET_FUSED=1 XVA_FP_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/p_on1 -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hifigan --no-xvapitch --no-fp32-parity --no-roofline > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_on1 -name "*.db" | head -1) $R/gpurun_out/onet_ab_1.csv > /dev/null
grep "onet_ln" $R/gpurun_out/onet_ab_1.csv | sed 's/(.*)"/"/'
cd $R; for m in 0 1 0 1; do XVA_FP_ONET_FUSED=$m python bench.py --steps 40 --warmup 5 --no-hifigan --no-xvapitch --no-fp32-parity --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('ONET_FUSED=$m', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; done
