#!/bin/bash
# stream-priority experiment: main chain (capture + replay stream) vs the side streams (weight gradients, geometry prefetch)
TAG=${1:-r02p}
mkdir -p gpurun_out
L=gpurun_out/prio_${TAG}.log
: > $L
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" >> $L 2>&1
run() { echo "=== $*" >> $L; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --skip-cpu-baseline --skip-roofline --skip-extras 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'])" >> $L 2>&1; }
run A=0
run M3D_MAIN_PRIO=-1
run M3D_MAIN_PRIO=-1 M3D_SIDE_PRIO=0
run M3D_MAIN_PRIO=0 M3D_SIDE_PRIO=-1
run M3D_SIDE_PRIO=-1
run A=0
for e in "A=0" "M3D_MAIN_PRIO=-1"; do
  echo "=== no-graph $e" >> $L
  env $e timeout 300 python bench.py --steps 20 --warmup 5 --no-graph --skip-cpu-baseline --skip-roofline --skip-extras 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'])" >> $L 2>&1
done
cat $L
