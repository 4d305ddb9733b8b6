#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r1'
# bench line (with cpu_baseline), rocprofv3 kernel stats of the same command, separate PMC passes.
set -u
R=${1:-r1}
O=gpurun_out/collect_$R
mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/final_bench_line.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.log 2>&1
for p in bf16x3 bf16; do python3 bench.py --precision $p --no-cpu-baseline > $O/${p}_bench_line.json 2>> $O/bench.err; done
python3 tools/bench_configs.py > $O/configs.jsonl 2>> $O/bench.err
python3 tools/taco_bench.py > $O/taco_b8.json 2>> $O/bench.err
python3 profiles/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write > $O/traffic.json
python3 profiles/summarize.py $(ls $O/stats/*/*kernel_trace.csv | head -1) > $O/by_grid.txt
python3 profiles/pmc_summarize.py $O/pmc_mfma > $O/pmc_mfma_by_kernel.txt
python3 profiles/pmc_summarize.py $O/pmc_fetch > $O/pmc_fetch_by_kernel.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_mfma
bash profiles/collect_small.sh $R > /dev/null 2>&1
tail -c 600 $O/final_bench_line.json
