# split-products FastPitch step: LayerNorm pair outputs (XVA_FP_LN_PAIRS) on / off, flash pairs on; parity tests of the mode first
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_fastpitch_gpu.py -q -k "split or planes or golden" 2>&1 | tail -5
for m in 0 1 0 1; do XVA_FP_LN_PAIRS=$m XVA_STEPS=20 python tools/fp_split_step.py 2>/dev/null | sed "s/$/ LN_PAIRS=$m/"; done
