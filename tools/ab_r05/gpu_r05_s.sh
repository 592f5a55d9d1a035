#!/bin/bash
# GPU box, round 5 call s: LDS-staged kNN walk for small clouds (deep levels, decoder 1-NN): bit-exact tests, per-level times,
# the step / eval forward against -DKNN_LDS=0
set -u
TAG=${1:-r05s}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
V=$ROOT/myria3d_amd/variants/libm3d_nolds.so
timeout -s KILL 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_pointnet2.py tests/test_gpu_predict.py -x -q -m gpu -k "knn or interpolate or eval or fps or predict or sa_" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $OUT/pytest_knn_$TAG.log
{ echo "== LDS-staged"; timeout -s KILL 200 python tools/knn_bench.py; echo "== global walk (KNN_LDS=0)"; M3D_LIB=$V timeout -s KILL 200 python tools/knn_bench.py; } 2>&1 | grep -v amdgpu.ids | tee $OUT/knn_lds_ab_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval forward', d['fwd_only']['ms_per_step'])"; }
for rep in 1 2; do
timeout -s KILL 300 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "LDS-staged "
M3D_LIB=$V timeout -s KILL 300 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "global walk"
done 2>&1 | tee $OUT/step_knn_lds_ab_$TAG.log
