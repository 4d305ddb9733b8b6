cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/c1voc; mkdir -p $O; cd $GRAFT_REPO_ROOT
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 tools/c1_voc_only.py > $O/st.log 2>&1
python3 profiles/summarize.py $(find $O/st -name '*kernel_trace.csv' | head -1) > $O/by_grid.txt
rm -rf $O/st
grep vocoder $O/st.log | tail -2; head -30 $O/by_grid.txt | cut -c1-150
