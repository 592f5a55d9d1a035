#!/bin/bash
# BatchNorm-backward-as-dgrad-prologue: parity, then A/B of the training step (M3D_FUSE_BN_DGRAD=0 = two-pass backward)
TAG=${1:-r02q}
mkdir -p gpurun_out
L=gpurun_out/bn_dgrad_${TAG}.log
: > $L
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_train.py -x -q -m gpu -k "layer or bn_dgrad or tail or train or grad or bf16" 2>&1 | tail -15 >> $L
run() { echo "=== $*" >> $L; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --skip-cpu-baseline --skip-roofline --skip-extras 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'])" >> $L 2>&1; }
run M3D_FUSE_BN_DGRAD=1
run M3D_FUSE_BN_DGRAD=0
run M3D_FUSE_BN_DGRAD=1
run M3D_FUSE_BN_DGRAD=0
grep -v amdgpu.ids $L
