R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_xvapitch_gpu.py -x -q -s -k "benchmarked_schedule" 2>&1 | grep -v Warn | grep "C5 full\|passed\|failed\|Error" | cut -c1-900
echo "== TEXT=split PITCH=split"; XVA_C5_TEXT_COMPUTE=split XVA_C5_PITCH_COMPUTE=split python bench.py --xvapitch-leg-only --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['parity']['loss_rel_by_name'], d['parity']['wave_rel'])"
