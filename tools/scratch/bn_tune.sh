#!/bin/bash
run() { env "$@" python bench.py --steps 40 --warmup 10 --launch graph --skip-cpu-baseline --skip-roofline --skip-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['fwd_only']['ms_per_step'])"; }
run A=0
run M3D_BN_APPLY_PER_THREAD=4 M3D_BN_APPLY_CAP=4096
run M3D_BN_APPLY_PER_THREAD=2 M3D_BN_APPLY_CAP=8192
run M3D_BN_APPLY_PER_THREAD=16 M3D_BN_APPLY_CAP=1024
run M3D_BN_RED_ROWS=4 M3D_BN_RED_CAP=4096
run M3D_BN_RED_ROWS=16 M3D_BN_RED_CAP=2048
run M3D_BN_RED_ROWS=4 M3D_BN_RED_CAP=8192
run A=0
