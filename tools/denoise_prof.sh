cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/dn; mkdir -p $O; cd $GRAFT_REPO_ROOT
python3 tools/denoise_bench.py 2>&1 | tail -4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 tools/denoise_bench.py > $O/st.log 2>&1
python3 profiles/summarize.py $(find $O/st -name '*kernel_trace.csv' | head -1) | grep -i "denoise\|overlap\|frame_counts\|total" | head
rm -rf $O/st
