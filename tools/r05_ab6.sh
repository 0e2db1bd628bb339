# A/B of the split-products FastPitch step: attention core as flash-style kernels on pairs (XVA_FP_ATT_FLASH=1) against the unfused chain (0); then a kernel trace of the flash form
R=$GRAFT_REPO_ROOT; cd $R
for m in 0 1 0 1; do XVA_FP_ATT_FLASH=$m XVA_STEPS=20 python tools/fp_split_step.py 2>/dev/null | sed "s/$/ ATT_FLASH=$m/"; done
cd /tmp && export TMPDIR=/tmp
XVA_SERIAL=1 XVA_STEPS=8 rocprofv3 --kernel-trace --stats -d /tmp/p_fl -o s -- python $R/tools/fp_split_step.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_fl -name "*.db" | head -1) $R/gpurun_out/fp_split_flash_stats.csv > /dev/null
head -40 $R/gpurun_out/fp_split_flash_stats.csv | cut -c1-150
