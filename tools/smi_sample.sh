#!/bin/bash
# Sample clocks / power with rocm-smi while a command runs:  tools/smi_sample.sh <out.txt> <cmd...>
out=$1; shift
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' ; echo; sleep 0.4; done ) > "$out" &
smi=$!
"$@"
rc=$?
kill $smi 2>/dev/null
exit $rc
