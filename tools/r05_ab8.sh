# split-products FastPitch step after batched guard zeroing and the planes split-K clamp: parity tests, step time, per-shape GEMM table
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_fastpitch_gpu.py tests/test_gemm_planes_gpu.py -q -k "split or planes or golden" 2>&1 | tail -5
for m in 1 1; do XVA_STEPS=20 python tools/fp_split_step.py 2>/dev/null; done
XVA_FP_MODE=split XVA_TOP=24 python tools/fp_gemm_profile.py 2>&1 | grep -v Warn | tail -26
