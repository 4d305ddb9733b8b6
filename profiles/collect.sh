#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r6'
# bench line (with cpu_baseline and the batch-1 / batch-8 sub-results), rocprofv3 kernel stats of the same command,
# FETCH_SIZE / WRITE_SIZE calibration on known-bytes kernels, separate PMC passes, the other precisions and configs.
set -u
R=${1:-r6}
O=gpurun_out/collect_$R
mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/final_bench_line.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-small --no-extra > $O/stats.log 2>&1
# one-stream schedule: per-launch durations that do not overlap (per-instantiation TFLOP/s table)
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --no-pipeline --steps 5 --warmup 2 --no-cpu-baseline --no-small --no-extra > $O/stats1.log 2>&1
# (the PMC passes below run the one-stream schedule: dispatch order = the library's launch log order)
# counter calibration: every kernel of tools/bin/traffic_calib moves exactly 1 GiB
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- tools/bin/traffic_calib > $O/cal.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- tools/bin/traffic_calib >> $O/cal.log 2>&1
python3 profiles/traffic_calib.py $O/cal_fetch $O/cal_write > $O/traffic_calib.json
rm -f $O/conv_log.csv
TTSAMD_HIFIGAN_STREAMS=0 TTSAMD_CONV_LOG=$O/conv_log.csv rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/pmc_fetch.log 2>&1
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/pmc_write.log 2>&1
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/pmc_mfma.log 2>&1
FRAMES=$(python3 -c "import json;print(json.load(open('$O/final_bench_line.json'))['config']['frames_per_step_rank0'])")
python3 profiles/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write $O/conv_log.csv $FRAMES $O/traffic_calib.json > $O/traffic.json
for p in bf16x3 bf16; do python3 bench.py --precision $p --no-cpu-baseline --no-extra > $O/${p}_bench_line.json 2>> $O/bench.err; done
python3 bench.py --precision bf16x3 --no-pipeline --no-cpu-baseline --no-small --no-extra > $O/bf16x3_one_stream_bench_line.json 2>> $O/bench.err
# config 3 inside the 1e-3 / 1e-4 tolerance (split bf16 on the octet engine, round 5): kernel stats (one stream), MFMA counters, per-layer table
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/xstats1 -- python3 bench.py --precision bf16x3 --no-pipeline --steps 5 --warmup 2 --no-cpu-baseline --no-small --no-extra > $O/xstats1.log 2>&1
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/xpmc_mfma -- python3 bench.py --precision bf16x3 --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/xpmc_mfma.log 2>&1
python3 profiles/summarize.py $(ls $O/xstats1/*/*kernel_trace.csv | head -1) > $O/bf16x3_by_grid_one_stream.txt
python3 profiles/pmc_summarize.py $O/xpmc_mfma > $O/bf16x3_pmc_mfma_by_kernel.txt
cp $(ls $O/xstats1/*/*kernel_stats.csv | head -1) $O/bf16x3_kernel_stats_one_stream.csv
python3 tools/bfo_bench.py --x3 --json $O/bfo3_layers.json > $O/bfo3_layers.txt 2>> $O/bench.err
rm -rf $O/xstats1 $O/xpmc_mfma
# fp32 same-box A/B behind DESIGN.md section 4: default routing (F(4,3) + F(2,3)), the round-5 routing (F(2,3) only: TTSAMD_WINO4=0 and the
# C = 64 pairs fused again), the direct kernels only
python3 bench.py --no-cpu-baseline --no-small --no-extra --steps 20 > $O/wino_on_bench_line.json 2>> $O/bench.err
TTSAMD_WINO4=0 TTSAMD_FUSED2_MASK=07f python3 bench.py --no-cpu-baseline --no-small --no-extra --steps 20 > $O/wino4_off_bench_line.json 2>> $O/bench.err
TTSAMD_WINO=0 TTSAMD_FUSED2_WB=0 TTSAMD_FUSED2_MASK=07f python3 bench.py --no-cpu-baseline --no-small --no-extra --steps 20 > $O/wino_off_bench_line.json 2>> $O/bench.err
# bf16 runs the two-stream schedule by default (FastPitch of step i+1 under HiFi-GAN of step i); the one-stream line of the same work:
python3 bench.py --precision bf16 --no-pipeline --no-cpu-baseline --no-small --no-extra > $O/bf16_one_stream_bench_line.json 2>> $O/bench.err
# config 3 (bf16 octet engine): kernel stats (three streams / one stream), HBM traffic and MFMA counters of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats -- python3 bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-small --no-extra > $O/bstats.log 2>&1
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats1 -- python3 bench.py --precision bf16 --no-pipeline --steps 5 --warmup 2 --no-cpu-baseline --no-small --no-extra > $O/bstats1.log 2>&1
rm -f $O/conv_log_bf16.csv
TTSAMD_HIFIGAN_STREAMS=0 TTSAMD_CONV_LOG=$O/conv_log_bf16.csv rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bpmc_fetch -- python3 bench.py --precision bf16 --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/bpmc_fetch.log 2>&1
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bpmc_write -- python3 bench.py --precision bf16 --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/bpmc_write.log 2>&1
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/bpmc_mfma -- python3 bench.py --precision bf16 --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/bpmc_mfma.log 2>&1
python3 profiles/traffic_from_pmc.py $O/bpmc_fetch $O/bpmc_write $O/conv_log_bf16.csv $FRAMES $O/traffic_calib.json > $O/traffic_bf16.json
python3 profiles/summarize.py $(ls $O/bstats/*/*kernel_trace.csv | head -1) > $O/bf16_by_grid.txt
python3 profiles/summarize.py $(ls $O/bstats1/*/*kernel_trace.csv | head -1) > $O/bf16_by_grid_one_stream.txt
python3 profiles/pmc_summarize.py $O/bpmc_mfma > $O/bf16_pmc_mfma_by_kernel.txt
python3 profiles/pmc_summarize.py $O/bpmc_fetch > $O/bf16_pmc_fetch_by_kernel.txt
cp $(ls $O/bstats1/*/*kernel_stats.csv | head -1) $O/bf16_kernel_stats_one_stream.csv
python3 tools/bfo_bench.py --json $O/bfo_layers.json > $O/bfo_layers.txt 2>> $O/bench.err
rm -rf $O/bstats $O/bstats1 $O/bpmc_fetch $O/bpmc_write $O/bpmc_mfma
python3 bench.py --gpus 2 --no-cpu-baseline > $O/dp2_one_device_bench_line.json 2>> $O/bench.err
python3 tools/taco_bench.py > $O/taco_b8.json 2>> $O/bench.err
# the full-size parity numbers (every utterance vs the oracle on the GPU: f32 / bf16x3 / bf16, B = 256 bf16 / bf16x3, configs 1, 4, 5)
python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -s 2>&1 | grep -E "full-size|config|passed|failed" > $O/fullsize_parity.txt
# round 4: the fused ResBlock pair kernels of the fp32 engine one launch at a time, the device-count probe, the torch-ROCm baseline (opt-in test)
python3 tools/fused_pair_bench.py > $O/fused_pair_layers.txt 2>> $O/bench.err
python3 tools/devcount_probe.py > $O/devcount_probe.txt 2>&1
TTSAMD_TORCH_GPU_BASELINE=1 python3 -m pytest tests/test_gpu_vs_torch_rocm.py -m gpu -q > $O/torch_rocm_baseline.log 2>&1; cp gpurun_out/torch_rocm_baseline.json $O/ 2>/dev/null
python3 profiles/summarize.py $(ls $O/stats/*/*kernel_trace.csv | head -1) > $O/by_grid.txt
python3 profiles/summarize.py $(ls $O/stats1/*/*kernel_trace.csv | head -1) > $O/by_grid_one_stream.txt
python3 profiles/pmc_summarize.py $O/pmc_mfma > $O/pmc_mfma_by_kernel.txt
python3 profiles/pmc_summarize.py $O/pmc_fetch > $O/pmc_fetch_by_kernel.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $O/stats1/*/*kernel_stats.csv | head -1) $O/kernel_stats_one_stream.csv
rm -rf $O/stats $O/stats1 $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/cal_fetch $O/cal_write
tail -c 600 $O/final_bench_line.json
