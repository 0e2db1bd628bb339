R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_fastpitch_gpu.py -x -q -k "golden or planes or ragged" 2>&1 | tail -6
for m in 0 1; do echo "== FFN_PLANES=$m"; XVA_FP_FFN_PLANES=$m python bench.py --steps 10 --warmup 3 --no-hifigan --no-xvapitch --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fastpitch_fp32_parity'])"; done
