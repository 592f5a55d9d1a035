#!/bin/bash
# GPU box: fp64 cross-lane sums by DPP / row swaps (GEMM statistics flush, wave_sum_d): full parity suite, GEMM timings, the step
set -u
timeout -s KILL 600 python -m pytest tests -m gpu -q --timeout 200 2>&1 | tail -3 | cut -c1-200
timeout -s KILL 120 python tools/opbench.py gemm 2>&1 | grep -E "TOTAL"
timeout -s KILL 200 python bench.py --steps 30 --warmup 8 --skip-cpu-baseline --skip-roofline --skip-extras 2>/dev/null | tail -1 | cut -c1-330
timeout -s KILL 200 python tools/graph_launch_probe.py 2>&1 | grep -E "alone|default stream:"
