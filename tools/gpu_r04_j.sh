#!/bin/bash
# GPU box, round 4 call J: the mlp2 / shortcut Linears of a block as one launch (m3d_gemm_pair_f32) — parity, then the
# step with and without it; fused step prologue / loss finalize / Adam tick; unfused BatchNorm backward for mlp_summit.
set -u
TAG=${1:-r04j}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $OUT/pytest_gpu_$TAG.log; cat $OUT/pytest_gpu_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
{
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"
M3D_PAIR_GEMMS=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph pair=0"
done
M3D_LIB=$V/libm3d_bp0.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph bp0"
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --precision bf16 2>/dev/null | tail -1 | step "graph bf16"
M3D_PAIR_GEMMS=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --precision bf16 2>/dev/null | tail -1 | step "graph bf16 pair=0"
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch eager 2>/dev/null | tail -1 | step "eager default"
} 2>&1 | tee $OUT/step_$TAG.log
bash tools/gpu_trace_analyze.sh $TAG > $OUT/trace_analyze_$TAG.log 2>&1; grep -E "^step:|per queue|main queue" $OUT/trace_analyze_$TAG.log
