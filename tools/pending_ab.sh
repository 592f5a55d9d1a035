#!/bin/bash
# Pending A/Bs (prepared at the end of round 4, no GPU minutes left; every switch is off in the stock build, whose code is
# byte-identical to the one without them).  (1) the LFA kernels with one-instruction maxima
# (-DM3D_FAST_MAX=1: v_med3_f32(a, b, +inf) instead of fmaxf's three v_max_f32) and the raw v_sqrt_f32 for the edge length
# (-DLFA_FAST_SQRT=1).  Static VALU count of lfa_fwd_kernel<8,16>: 500 -> 457, lfa_bwd_kernel<8,16>: 581 -> 552; the default build
# is byte-identical to the one without the switches.  (2) -DLFA_RED_WIDE=1: eight partials in flight in both loops of the LFA
# partial-sum reduce (its G sums are ONE load per trip: 73 dependent round trips per thread at ch = 256, the critical path of the
# 72 us launch).  (3) -DROWS_GATHER_BATCH=1: gather_sum_rows with eight contributors' ids, then rows, per trip.  Same sums in the
# same order in (2) and (3).  The variant library carries all three; split them if the step moves.
#   here (no GPU):  tools/pending_ab.sh build      -> myria3d_amd/variants/libm3d_lfadiet.so
#   on the GPU box: tools/pending_ab.sh run [TAG]  -> parity of the variant, per-level kernel times, step, both libraries
set -eu
ROOT=$(cd $(dirname $0)/.. && pwd)
if [ "${1:-build}" = build ]; then
  cd $ROOT/myria3d_amd/csrc; make > /dev/null; mkdir -p ../variants
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-variable -DM3D_FAST_MAX=1 -DLFA_FAST_SQRT=1 -DLFA_RED_WIDE=1 -DROWS_GATHER_BATCH=1"
  /opt/rocm/bin/hipcc $FL -c lfa.hip -o /tmp/var_lfadiet_fwd.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc $FL -c lfa_bwd.hip -o /tmp/var_lfadiet_bwd.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc $FL -c rows.hip -o /tmp/var_lfadiet_rows.o 2>&1 | grep -E "error" || true
  OBJS=$(ls *.o | grep -v "^lfa.o$" | grep -v "^lfa_bwd.o$" | grep -v "^rows.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/var_lfadiet_fwd.o /tmp/var_lfadiet_bwd.o /tmp/var_lfadiet_rows.o -o ../variants/libm3d_lfadiet.so
  echo built myria3d_amd/variants/libm3d_lfadiet.so
  exit 0
fi
TAG=${2:-lfadiet}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
V=$ROOT/myria3d_amd/variants/libm3d_lfadiet.so
cd $ROOT
M3D_LIB=$V timeout -s KILL 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -x -q -k "lfa or train or golden or reference or parity or csr or gather" 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $OUT/pytest_$TAG.log
{ echo "== default"; timeout -s KILL 200 python tools/opbench.py lfa | grep "^lfa"; echo "== variant"; M3D_LIB=$V timeout -s KILL 200 python tools/opbench.py lfa | grep "^lfa"; } > $OUT/lfa_opbench_$TAG.log 2>&1; cat $OUT/lfa_opbench_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"
M3D_LIB=$V timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph variant"
done 2>&1 | tee $OUT/step_$TAG.log
