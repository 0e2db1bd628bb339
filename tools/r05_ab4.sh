R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
XVA_SERIAL=1 rocprofv3 --kernel-trace --stats -d /tmp/p_sp -o s -- python $R/tools/fp_split_step.py > /tmp/sp.txt 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_sp -name "*.db" | head -1) $R/gpurun_out/fp_split_stats.csv > /dev/null
head -32 $R/gpurun_out/fp_split_stats.csv | cut -c1-150
