#!/bin/bash
# same-box A/B of environment switches inside the bench step, ALTERNATED (box drift and run-to-run noise are 0.2-0.5 % of a step, so a
# single pair decides nothing):   gpurun -- 'bash tools/ab_env.sh "TTSAMD_XCD_W=0" "TTSAMD_COMPACT=0"'
#   AB_ROUNDS=5 (default 3) rounds of: baseline, then every variant; prints every run and min / median / max per variant.
#   AB_ARGS="--precision bf16x3" adds bench.py arguments;  AB_STEPS (20)
R=${AB_ROUNDS:-3}
S=${AB_STEPS:-20}
T=$(mktemp -d)
run() { env $1 python3 bench.py --no-cpu-baseline --no-small --no-extra --steps $S --warmup 4 ${AB_ARGS:-} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; }
for r in $(seq $R); do
  i=0
  for e in "TTSAMD_NOP=1" "$@"; do
    out=$(run "$e"); echo "round $r  $e: $out"
    echo "$out" >> $T/$i; i=$((i + 1))
  done
done
i=0
for e in "TTSAMD_NOP=1" "$@"; do
  python3 - "$e" $T/$i <<'PY'
import sys, statistics
v = sorted(float(l.split()[0]) for l in open(sys.argv[2]) if l.strip())
print('%-40s ms/step min %.3f  median %.3f  max %.3f  (n = %d, spread %.2f %%)' % (sys.argv[1] if sys.argv[1] != 'TTSAMD_NOP=1' else 'baseline', v[0], statistics.median(v), v[-1], len(v), 100 * (v[-1] - v[0]) / v[0]))
PY
  i=$((i + 1))
done
rm -rf $T
