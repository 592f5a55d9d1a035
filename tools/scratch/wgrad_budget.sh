#!/bin/bash
for cfg in "6144 12288" "4096 8192" "8192 16384" "3072 6144" "100000 100000"; do
  set -- $cfg
  M3D_WGRAD_BATCH_WAVES_BIG=$1 M3D_WGRAD_BATCH_WAVES_SMALL=$2 python bench.py --steps 40 --warmup 10 --launch graph --skip-cpu-baseline --skip-roofline --skip-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('budget $1 $2:', d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_net.py -x -q -m gpu 2>&1 | tail -2
