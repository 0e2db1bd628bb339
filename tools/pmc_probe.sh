cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/pp$i -o c -- python $R/tools/gemm_pmc_probe.py 6 > /tmp/pp$i.log 2>&1 || tail -3 /tmp/pp$i.log
  python $R/tools/pmc_summary.py $(find /tmp/pp$i -name "*.db" | head -1) $R/gpurun_out/pmc_probe_$i.csv
done
grep -h "glds" $R/gpurun_out/pmc_probe_*.csv | sed 's/void xva_glds:://; s/(xva_gemm_params, int)//' | sort
