#!/bin/bash
# HBM traffic of the fp32 step only (the FETCH_SIZE / WRITE_SIZE passes of collect.sh):  gpurun -- 'bash profiles/collect_traffic.sh r6'
set -u
R=${1:-r6}
O=gpurun_out/traffic_$R
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- tools/bin/traffic_calib > $O/cal.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- tools/bin/traffic_calib >> $O/cal.log 2>&1
python3 profiles/traffic_calib.py $O/cal_fetch $O/cal_write > $O/traffic_calib.json
rm -f $O/conv_log.csv
TTSAMD_HIFIGAN_STREAMS=0 TTSAMD_CONV_LOG=$O/conv_log.csv rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/pmc_fetch.log 2>&1
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/pmc_write.log 2>&1
FRAMES=$(python3 -c "import json;print([json.loads(l) for l in open('$O/pmc_fetch.log') if l.startswith('{')][-1]['config']['frames_per_step_rank0'])")
python3 profiles/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write $O/conv_log.csv $FRAMES $O/traffic_calib.json > $O/traffic.json
rm -rf $O/pmc_fetch $O/pmc_write $O/cal_fetch $O/cal_write
python3 -c "import json; t=json.load(open('$O/traffic.json')); print('launches', t['conv_launches'], 'measured / algorithmic per launch', round(t['bytes_per_conv_launch_corrected']/1e6,1), round(t['algorithmic_bytes_per_conv_launch']/1e6,1), 'ratio', round(t['ratio'],3))"
