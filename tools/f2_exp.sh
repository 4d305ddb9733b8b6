# Timing experiments on the second-generation fused fp32 pair (resblock_fused2.hip, -DTTS_F2_EXP hooks: no MFMAs / no loads /
# late second block ...).  The hooks give WRONG results by design, so this script builds ITS OWN library under /tmp and points
# the binding at it (TTSAMD_LIB); the product library in tts-arabic-pytorch_amd/ttsamd/lib is never touched.
#   gpurun -- 'bash tools/f2_exp.sh'
export TMPDIR=/tmp
set -e
X=/tmp/ttsamd_f2exp
make -C tts-arabic-pytorch_amd/csrc OBJ=$X/obj OUT=$X EXTRA="-DTTS_F2_EXP -DTTS_EXPERIMENT" -j8 > $X.build.log 2>&1
export TTSAMD_LIB=$X/libttsamd.so
set +e
O=gpurun_out/f2_exp
mkdir -p $O
for e in 0 2 4 6 8 $((16+256*10)) $((16+256*20)) $((16+256*40)) $((16+256*80)); do echo "== TTSAMD_F2_EXP=$e"; TTSAMD_F2_EXP=$e python3 tools/fused_pair_bench.py --cases 32:3:5,64:3:5,64:11:5,128:3:5 --variants 2,3 --reps 10 2>&1 | grep "C="; done > $O/exp2.txt
cat $O/exp2.txt
# PMC pass of the same launches (what produced profiles/r4/pmc_fused.txt).  The interpreter goes directly after `--`: never the script
# itself, `env`, or a shell -- the profiler's preloaded runtime has initialised the GPU by then and such a hop is a forbidden exec.
TTSAMD_F2_EXP=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python3 tools/fused_pair_bench.py --cases 32:3:5,64:3:5,64:11:5,128:3:5 --variants 2,3 --reps 3 > $O/pmc.log 2>&1
python3 profiles/pmc_summarize.py $O/pmc > $O/pmc_fused.txt
rm -rf $O/pmc
