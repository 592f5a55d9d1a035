#!/bin/bash
# GPU box, round 4 call Q: the LFA partial-sum reduce alone, per layer shape, current build against the round-3 kernel.
set -u
TAG=${1:-r04q}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
{
echo "== current"; timeout -s KILL 200 python tools/lfa_reduce_bench.py
for v in redold $(ls $V | sed -n 's/libm3d_\(red[a-z0-9]*\)\.so/\1/p' | grep -v redold); do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python tools/lfa_reduce_bench.py; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/lfa_reduce_bench_$TAG.log
