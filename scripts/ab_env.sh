#!/bin/bash
# A/B runtime environment settings on the same box: ab_env.sh "VAR=1 VAR2=3" "" ...  (empty string = defaults)
# ROUNDS (default 2) passes over the list, STEPS (default 20) timed steps per run.
ROUNDS=${ROUNDS:-2}; STEPS=${STEPS:-20}
for round in $(seq 1 $ROUNDS); do
  for E in "$@"; do
    env $E python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('[$E] round $round:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; conv family', d['roofline']['kernel_ms_per_step'], 'ms')"
  done
done
