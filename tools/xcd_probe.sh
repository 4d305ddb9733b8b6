#!/bin/bash
# stand-alone: the C = 256 and C = 128 k = 11 convs of the bench step with the co-tiles of a time tile spread over the XCDs (WMAX 0) or on one XCD,
# time per launch and FETCH_SIZE per launch (own PMC pass)
O=gpurun_out/xcd_probe; mkdir -p $O
export PROD=1 RES_SEP=1 RAGGED=auto CUSTOM="32,256,3,1,3584;32,256,7,3,3584;32,256,11,5,3584;32,128,11,5,28672;32,128,7,3,28672"
for w in ${TIMEW:-0 900 2000 3000}; do echo "== TTSAMD_XCD_WMAX_KB=$w"; TTSAMD_XCD_WMAX_KB=$w tools/bin/conv_bench; done
for w in ${PMCW:-0 3000}; do
  export TTSAMD_XCD_WMAX_KB=$w
  ITERS=2 WARM_MS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$w -- tools/bin/conv_bench > /dev/null 2>&1
  python3 - $O/fetch_$w $w <<'PY'
import csv, glob, sys, collections
d = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE': continue
        k = (r['Kernel_Name'][:60], r['Grid_Size'])
        d.setdefault(k, []).append(float(r['Counter_Value']))
for k, v in d.items():
    print('WMAX', sys.argv[2], k, 'launches', len(v), 'FETCH_SIZE x2 MB/launch %.1f' % (2 * sum(v) / len(v) * 1024 / 1e6 / 1.0))
PY
done
