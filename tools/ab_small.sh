run() { echo -n "$1: "; env $1 python3 bench.py --batch ${BATCH:-1} --no-cpu-baseline --no-small --no-extra --steps 40 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms/step' % (d['ms_per_step']))"; }
run "TTSAMD_NOP=1"
for e in "$@"; do run "$e"; done
run "TTSAMD_NOP=1"
