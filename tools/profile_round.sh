#!/bin/bash
# Round-end evidence on the GPU box: plain bench line, rocprofv3 kernel stats of the same command, and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs as MI355X_MICROARCH.md prescribes) per leg.  Everything lands in gpurun_out/final/.
# usage (from the build container):  gpurun -- 'XVA_COMMIT=<short sha> XVA_ROUND=r04 bash tools/profile_round.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
RD=${XVA_ROUND:-r06}
O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export XVA_BENCH_C5_INPROCESS=1   # the rocprofv3 runs of bench.py keep the xVAPitch leg in the traced process (one database); the plain reference line below unsets it
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f16 --no-trainer-leg 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${RD}_final_bench_under_rocprof.json
python $R/tools/rocpd_summary.py $(find /tmp/p_stats -name "*.db" | head -1) $O/${RD}_final_bench_kernel_stats.csv
# the same two legs with the engines' stream lanes off (XVA_*_STREAMS=1): kernels do not overlap, so the per-kernel average durations are the
# kernels' own — the numbers bench.py's roofline passes (lanes off as well) have to agree with
XVA_FP_STREAMS=1 XVA_HG_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/p_ser -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f16 --no-trainer-leg 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${RD}_serial_lanes_bench_under_rocprof.json
python $R/tools/rocpd_summary.py $(find /tmp/p_ser -name "*.db" | head -1) $O/${RD}_serial_lanes_kernel_stats.csv
XVA_FP_STREAMS=1 XVA_HG_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/p_fps -o s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f16 --no-trainer-leg --no-hifigan --no-xvapitch --no-fp32-parity 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${RD}_fastpitch_only_serial_lanes_under_rocprof.json
python $R/tools/rocpd_summary.py $(find /tmp/p_fps -name "*.db" | head -1) $O/${RD}_fastpitch_only_serial_lanes_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_fp -o f -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f16 --no-trainer-leg --no-hifigan --no-xvapitch --no-fp32-parity 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${RD}_fastpitch_only_under_rocprof.json
python $R/tools/rocpd_summary.py $(find /tmp/p_fp -name "*.db" | head -1) $O/${RD}_fastpitch_only_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_hg -o h -- python $R/tools/hg_phase_timing.py > $O/${RD}_hifigan_phase_timing.txt 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/p_hg -name "*.db" | head -1) $O/${RD}_hifigan_only_kernel_stats.csv
# the xVAPitch C5 iteration: per-pass timing + per-shape xva_gemm table (plain run), kernel stats of the same command under rocprofv3
XVA_C5_GEMM_PROFILE=1 python $R/tools/c5_step_time.py 16 100 400 bf16 bf16 2>/dev/null | grep -v Warning > $O/${RD}_xvapitch_c5_timing.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o c -- python $R/tools/c5_step_time.py 16 100 400 bf16 bf16 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_c5 -name "*.db" | head -1) $O/${RD}_xvapitch_c5_kernel_stats.csv
# occupancy timelines of the three steps from the kernel traces above (GPU busy union, idle gaps, kernels in flight, who runs alone): tools/trace_gaps.py
rocprofv3 --kernel-trace -d /tmp/p_fpt -o t -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f16 --no-trainer-leg --no-hifigan --no-xvapitch --no-fp32-parity --no-roofline > /dev/null 2>&1   # no roofline passes: their LAMB timing loops would be the last markers
python $R/tools/trace_dump.py $(find /tmp/p_fpt -name "*.db" | head -1) /tmp/fp_trace.csv > /dev/null && python $R/tools/trace_gaps.py /tmp/fp_trace.csv lamb_pass1 5 > $O/${RD}_fastpitch_timeline.txt 2>&1
python $R/tools/trace_dump.py $(find /tmp/p_hg -name "*.db" | head -1) /tmp/hg_trace.csv > /dev/null && python $R/tools/trace_gaps.py /tmp/hg_trace.csv adamw_kernel 6 > $O/${RD}_hifigan_timeline.txt 2>&1
python $R/tools/trace_dump.py $(find /tmp/p_c5 -name "*.db" | head -1) /tmp/c5_trace.csv > /dev/null && python $R/tools/trace_gaps.py /tmp/c5_trace.csv adamw_kernel 4 > $O/${RD}_xvapitch_c5_timeline.txt 2>&1
# when the xVAPitch generator group's buckets can start their exchange (one rank, marker kernels in place of the all-reduce)
rocprofv3 --kernel-trace -d /tmp/p_c5dp -o d -- python $R/tools/dp_overlap_probe_c5.py > /dev/null 2>&1
python $R/tools/trace_dump.py $(find /tmp/p_c5dp -name "*.db" | head -1) /tmp/c5dp_trace.csv > /dev/null && python $R/tools/dp_overlap_probe_c5.py --report /tmp/c5dp_trace.csv > $O/${RD}_dp_overlap_probe_c5.txt 2>&1
# the mel front end's kernels per mode (fused kernel / four-launch FFT pipeline / dense DFT)
rocprofv3 --kernel-trace --stats -d /tmp/p_mel -o m -- python $R/tools/mel_time.py > $O/${RD}_mel_front_end.txt 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/p_mel -name "*.db" | head -1) /tmp/mel_stats.csv > /dev/null && grep -i "mel\|stft\|magnitude\|reflect\|gemm" /tmp/mel_stats.csv >> $O/${RD}_mel_front_end.txt
for leg in fastpitch hifigan; do
  if [ $leg = fastpitch ]; then CMD="python $R/bench.py --steps 2 --warmup 1 --no-hifigan --no-cpu-baseline --no-f16 --no-trainer-leg --no-roofline --no-xvapitch --no-fp32-parity"; else CMD="python $R/tools/hg_gemm_profile.py 64"; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    XVA_FP_STREAMS=1 XVA_HG_STREAMS=1 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/p_${leg}_$ctr -o c -- $CMD > /dev/null 2>&1
    python $R/tools/pmc_summary.py $(find /tmp/p_${leg}_$ctr -name "*.db" | head -1) /tmp/${leg}_$ctr.csv
  done
  python $R/tools/pmc_merge.py /tmp/${leg}_FETCH_SIZE.csv /tmp/${leg}_WRITE_SIZE.csv $O/${RD}_${leg}_pmc_hbm_bytes.csv
  python - <<PY
import json, sys
sys.path.insert(0, "$R")
import bench
json.dump({"csrc": bench.csrc_fingerprint(), "commit": "${XVA_COMMIT:-unknown}", "counters": "FETCH_SIZE (KB, x2 gfx950 correction applied by the reader) and WRITE_SIZE (KB), separate rocprofv3 --pmc passes"},
          open("$O/${RD}_${leg}_pmc_hbm_bytes.meta.json", "w"))
PY
done
# round 5 evidence: the D-step backward per launch, the resident-input convolution against the staggered general tiles on the generator's
# 128 / 256-channel shapes (item 2b), LayerNorm forms, the split-products FastPitch step per kernel (item 4), split-K by atomics (item 2c)
XVA_HG_PHASE3=1 python $R/tools/hg_gemm_profile.py 64 > $O/${RD}_hifigan_gemm_profile.txt 2>/dev/null
for m in -1 7 6; do echo "== xva_gemm_set_mainloop($m): -1 automatic (resident-input kernel), 7 forced 384x128 staggered tile, 6 automatic without the resident-input kernel"; XVA_BENCH_C=128 python $R/tools/conv_res_bench.py $m 2>/dev/null | tail -6; done > $O/${RD}_conv_res_tile_modes.txt
python $R/tools/ln_time.py > $O/${RD}_layernorm_timing.txt 2>/dev/null
python $R/tools/fp_split_step.py 2>/dev/null | tail -1 > $O/${RD}_fastpitch_split_step.txt
XVA_FP_FFN_PLANES=0 python $R/tools/fp_split_step.py 2>/dev/null | tail -1 >> $O/${RD}_fastpitch_split_step.txt
XVA_SERIAL=1 rocprofv3 --kernel-trace --stats -d /tmp/p_sp -o s -- python $R/tools/fp_split_step.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_sp -name "*.db" | head -1) $O/${RD}_fastpitch_split_kernel_stats.csv
# the lanes-off FastPitch kernel trace gets the same fingerprint: bench.py quotes roofline.frac_rocprof from it only while it describes the sources it runs
cp $O/${RD}_fastpitch_pmc_hbm_bytes.meta.json $O/${RD}_fastpitch_only_serial_lanes_kernel_stats.meta.json
# the plain bench line last: its roofline.traffic / frac_rocprof read the tables just measured (same sources: fingerprint checked)
cp $O/${RD}_*_pmc_hbm_bytes.csv $O/${RD}_*_pmc_hbm_bytes.meta.json $O/${RD}_fastpitch_only_serial_lanes_kernel_stats.csv $O/${RD}_fastpitch_only_serial_lanes_kernel_stats.meta.json $R/profiles/
cd $R && XVA_BENCH_C5_INPROCESS=0 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/${RD}_final_bench.json
cp $R/bench_detail.json $O/${RD}_final_bench_detail.json   # the tables / notes behind the compact line
ls -la $O
# the fused ResBlock pair against the two launches (time, per-kernel, PMC traffic) and the run-to-run reproducibility of the three engines on their stream lanes
python $R/tools/determinism_sweep.py 5 2>/dev/null | grep -v amdgpu.ids > $O/${RD}_determinism_sweep_raw.txt
# round 6 evidence: the vendor-GEMM yardstick, the fp16-operand FastPitch step per kernel with the lanes off (item 1), the trainers' own meters
# from a dataset directory with and without the prefetcher (item 3), the 256 x 256 workgroup's phase timeline
python $R/tools/gemm_yardstick.py 20 2>/dev/null > $O/${RD}_gemm_yardstick.txt
XVA_FP_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/p_f16 -o s -- python $R/tools/fp_step_time.py 20 f16 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_f16 -name "*.db" | head -1) $O/${RD}_fastpitch_f16_serial_lanes_kernel_stats.csv
for m in bf16 f16; do python $R/tools/fp_step_time.py 30 $m 2>/dev/null | tail -1; done > $O/${RD}_fastpitch_step_times.txt
cd $R && python bench.py --trainer-leg 2>/dev/null | tail -1 > $O/${RD}_trainer_leg_prefetch.json
cd $R && XVA_PREFETCH=0 python bench.py --trainer-leg 2>/dev/null | tail -1 > $O/${RD}_trainer_leg_no_prefetch.json
mkdir -p $R/build && hipcc --offload-arch=gfx950 -O3 -std=c++17 -DXVA_GLDS_TIMING -I$R/xva-trainer_amd/csrc -I$R/include $R/tools/glds_timing.hip $R/xva-trainer_amd/csrc/core.hip -o $R/build/glds_timing 2>/dev/null && $R/build/glds_timing 0 > $O/${RD}_glds_phase_timing.txt
# later round-6 probes: producer -> consumer through the Infinity Cache (item 2c), the attention block's K <= 192 products per tile, the fused fp16-mode tail,
# every xva_gemm launch of one FastPitch fwd+bwd per shape in both modes (lanes off)
python $R/tools/producer_consumer_probe.py 2>/dev/null > $O/${RD}_producer_consumer_probe_raw.txt
python $R/tools/thin_gemm_ab.py 2>/dev/null > $O/${RD}_thin_gemm_tiles.txt
python $R/tools/onet_f16_time.py 2>/dev/null > $O/${RD}_onet_f16_fused.txt
cd $R && XVA_FP_STREAMS=1 XVA_TOP=60 python tools/fp_gemm_profile.py 2>/dev/null > $O/${RD}_fastpitch_gemm_profile_bf16.txt
cd $R && XVA_FP_STREAMS=1 XVA_TOP=70 XVA_FP_MODE=f16 python tools/fp_gemm_profile.py 2>/dev/null > $O/${RD}_fastpitch_gemm_profile_f16.txt
# global -> LDS DMA by gather shape (half lines / whole lines / half-line pairs back to back), and the staggered K loop's ablations
hipcc --offload-arch=gfx950 -O3 -w $R/tools/dma_pattern_probe.hip -o $R/build/dma_probe 2>/dev/null && $R/build/dma_probe > $O/${RD}_dma_pattern_probe.txt
hipcc --offload-arch=gfx950 -O3 -w $R/tools/dma_issue_probe.hip -o $R/build/dma_issue 2>/dev/null && $R/build/dma_issue > $O/${RD}_dma_issue_probe.txt
for a in 0 1 2 3 32; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DXVA_GLDS_TIMING -DXVA_GLDS_ABLATE=$a -I$R/xva-trainer_amd/csrc -I$R/include $R/tools/glds_timing.hip $R/xva-trainer_amd/csrc/core.hip -o $R/build/glds_timing_a 2>/dev/null && echo "XVA_GLDS_ABLATE=$a (1: no MFMAs, 2: no DMA, 32: vmcnt waits one tile looser)" && $R/build/glds_timing_a 0 | grep "staggered blocks   648"; done > $O/${RD}_kloop_ablation_raw.txt
