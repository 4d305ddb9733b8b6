#!/bin/bash
# activation window as non-temporal loads (tools/bin/conv_bench_nt = conv_bench built with -DTTS_X_NT) vs default: time and FETCH_SIZE per launch
O=gpurun_out/nt_probe; mkdir -p $O
export PROD=1 RES_SEP=1 RAGGED=auto CUSTOM="32,256,7,3,3584;32,256,11,5,3584;32,128,11,5,28672;32,128,7,3,28672;32,128,3,1,28672"
for t in conv_bench conv_bench_nt; do echo "== $t"; tools/bin/$t; done
for t in conv_bench conv_bench_nt; do
  ITERS=2 WARM_MS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$t -- tools/bin/$t > /dev/null 2>&1
  python3 - $O/$t $t <<'PY'
import csv, glob, sys, collections
d = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE' or 'conv1d' not in r['Kernel_Name']: continue
        d.setdefault((r['Kernel_Name'][:60], r['Grid_Size']), []).append(float(r['Counter_Value']))
for k, v in d.items():
    print(sys.argv[2], k, 'FETCH_SIZE x2 MB/launch %.1f' % (2 * sum(v) / len(v) * 1024 / 1e6))
PY
done
