R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gemm_planes_gpu.py tests/test_fastpitch_gpu.py -x -q -k "planes or split3" 2>&1 | tail -3
python tools/fp_split_step.py 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
XVA_SERIAL=1 rocprofv3 --kernel-trace --stats -d /tmp/p_sp -o s -- python $R/tools/fp_split_step.py > /tmp/sp.txt 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_sp -name "*.db" | head -1) $R/gpurun_out/fp_split_stats.csv > /dev/null
head -12 $R/gpurun_out/fp_split_stats.csv | cut -c1-130
