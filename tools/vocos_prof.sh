cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/vocos; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 tools/vocos_bench.py > $O/st.log 2>&1
python3 profiles/summarize.py $(find $O/st -name '*kernel_trace.csv' | head -1) > $O/by_grid.txt
rm -rf $O/st
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python3 tools/vocos_bench.py > $O/pmc.log 2>&1
python3 profiles/pmc_summarize.py $O/pmc > $O/pmc_mfma_by_kernel.txt
rm -rf $O/pmc
