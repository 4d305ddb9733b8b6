#!/bin/bash
# the two PMC passes (FETCH_SIZE, WRITE_SIZE) of one fp32 bench step + the launch log -> measured vs algorithmic HBM bytes per conv launch
# (the part of profiles/collect.sh that produces traffic.json), then the step time
O=gpurun_out/traffic_pass; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- tools/bin/traffic_calib > $O/cal.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- tools/bin/traffic_calib >> $O/cal.log 2>&1
python3 profiles/traffic_calib.py $O/cal_fetch $O/cal_write > $O/traffic_calib.json
TTSAMD_HIFIGAN_STREAMS=0 TTSAMD_CONV_LOG=$O/conv_log.csv rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/pmc_fetch.log 2>&1
TTSAMD_HIFIGAN_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --no-pipeline --steps 1 --warmup 1 --no-cpu-baseline --no-small --no-extra > $O/pmc_write.log 2>&1
python3 profiles/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write $O/conv_log.csv ${FRAMES:-14341} $O/traffic_calib.json > $O/traffic.json
python3 profiles/pmc_summarize.py $O/pmc_fetch > $O/pmc_fetch_by_kernel.txt
python3 -c "import json; t=json.load(open('$O/traffic.json')); print('traffic', t['bytes_per_conv_launch_corrected'], t['algorithmic_bytes_per_conv_launch'], t['ratio'])"
rm -rf $O/cal_fetch $O/cal_write $O/pmc_fetch $O/pmc_write
for i in 1 2; do python3 bench.py --no-cpu-baseline --no-small --no-extra --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('step', d['ms_per_step'], d['roofline']['frac'])"; done
