"""GPU box (under rocprofv3 --kernel-trace): ONE predict_cloud call on bench.py's 10 M-point cloud after a warm-up call;
tools/gpu_r05_g.sh turns the trace into per-queue busy time and the gaps of the main queue."""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(bench.predict_e2e_bench(argparse.Namespace(), torch.device("cuda:0"), reps=1))
