#!/bin/bash
# knock-out timing (WRONG numerics, timing only): how much of the step does a kernel family cost on the wall clock?
for k in none wgrad bnreduce none; do
  for m in "--launch graph" "--launch eager"; do
    M3D_KNOCKOUT=$k python bench.py --steps 40 --warmup 45 $m --skip-cpu-baseline --skip-roofline --skip-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', '$m', d['ms_per_step'])"
  done
done
