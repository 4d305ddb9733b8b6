export TMPDIR=/tmp
O=gpurun_out/w4prof; mkdir -p $O
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --no-pipeline --steps 5 --warmup 2 --no-cpu-baseline --no-small --no-extra > $O/st.log 2>&1
python3 profiles/summarize.py $(find $O/st -name '*kernel_trace.csv' | head -1) > $O/by_grid_one_stream.txt
rm -rf $O/st
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python3 bench.py --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/pmc.log 2>&1
python3 profiles/pmc_summarize.py $O/pmc > $O/pmc_mfma_by_kernel.txt
rm -rf $O/pmc
