R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/t_fp -o f -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-hifigan --no-xvapitch --no-fp32-parity --no-roofline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/fp_line.json
python $R/tools/trace_dump.py $(find /tmp/t_fp -name "*.db" | head -1) $O/fp_trace.csv
rocprofv3 --kernel-trace -d /tmp/t_hg -o h -- python $R/tools/hg_phase_timing.py > $O/hg_phase.txt 2>/dev/null
python $R/tools/trace_dump.py $(find /tmp/t_hg -name "*.db" | head -1) $O/hg_trace.csv
rocprofv3 --kernel-trace -d /tmp/t_c5 -o c -- python $R/tools/c5_step_time.py 16 100 400 bf16 bf16 > $O/c5.txt 2>/dev/null
python $R/tools/trace_dump.py $(find /tmp/t_c5 -name "*.db" | head -1) $O/c5_trace.csv
gzip -f $O/*.csv
ls -la $O
