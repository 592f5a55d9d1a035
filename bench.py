#!/usr/bin/env python
"""Benchmark of the RandLA-Net hot path on MI355X (BASELINE.json metric: points/sec fwd+bwd on 12 800-pt tiles).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller launches it under ``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`` (the driver does), or — when WORLD_SIZE is not in the environment —
``python bench.py --gpus N`` re-executes itself under torch.distributed.run with N ranks.  The world size that comes
up must equal ``--gpus`` (and fit ``torch.cuda.device_count()``), otherwise the run aborts.

One "step" = one pass of the hot path over one batch of synthetic tiles already resident in HBM:
train-mode forward (BatchNorm batch statistics, dropout, device-side random decimation) + cross-entropy +
backward + (N>1: ONE flat 4.45 MB gradient all-reduce over RCCL) + Adam update.  Workload = BASELINE config 2:
16 tiles x 12 800 points per GPU, K=16, F=9, C=6 (weak scaling: every rank owns its own 16 tiles; tiles are
independent units, the gradient all-reduce is the only collective).  Arithmetic is fp32 (the reference's dtype).

Rank 0 prints ONE JSON line with the contract keys plus
  "fwd_only": eval-mode forward throughput of the same batch,
  "roofline": achieved vs peak for the dominant kernel (LFA backward, fp32 MFMA), timed live with HIP events on the
              launch stream; "roofline_knn_lse_stage": the HBM-class kernels of the kNN + LSE gather at level 1,
  "cpu_baseline": the CPU oracle (restated reference path, torch + cKDTree) timed on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tiles", type=int, default=16, help="tiles per GPU")
    ap.add_argument("--points", type=int, default=12800, help="points per tile")
    ap.add_argument("--neighbors", type=int, default=16)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph (= --launch eager)")
    ap.add_argument("--launch", choices=("auto", "graph", "eager"), default="auto",
                    help="how the timed steps are launched: replayed hipGraph, kernel by kernel, or (default) whichever a "
                         "10-step probe of each measures faster on this host — both run the same kernels; since the host "
                         "path was trimmed the eager step is no longer launch-bound (DESIGN.md section 4)")
    ap.add_argument("--no-lookahead", dest="lookahead", action="store_false",
                    help="build the position-only tables (kNN, decimation) of a step inside that step instead of one "
                         "step ahead (HipRandLANet.prefetch_geometry; bit-identical results either way)")
    ap.add_argument("--lookahead", dest="lookahead", action="store_true", help="(default)")
    ap.set_defaults(lookahead=True)
    ap.add_argument("--lookahead-mode", choices=("dual", "single"), default="dual",
                    help="hipGraph launch of the lookahead: 'dual' (default) = the step and the next step's position-only "
                         "work as two graphs replayed on two streams, so that they overlap; 'single' = one graph holding both "
                         "branches (the executor then runs the position-only branch first, ~0.8 ms ahead of the features). "
                         "dual was the slower one while the weight gradients still ran on a side stream (profiles/r02vwx_*), "
                         "and is 0.08 ms faster since they are batched at the end of the backward pass")
    ap.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                    help="matmul precision of the timed net (the contract line is fp32; 'bf16' is what the \"bf16\" leg runs)")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-roofline", action="store_true")
    ap.add_argument("--skip-extras", action="store_true",
                    help="skip the informative extra legs of the N=1 line (eager step, predict sweep, dense tiles)")
    ap.add_argument("--cpu-tiles", type=int, default=16, help="tiles in the CPU-baseline sample (BASELINE.md 3: 16)")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="BASELINE.md 3 protocol in full: 3 warm-up + 10 timed iterations, at the probe-picked thread "
                         "count AND on all cores (several minutes of CPU time)")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="launch check without GPUs: bring the N ranks up over gloo, exchange one all-reduce, print the "
                         "rank census (used by the CPU tests)")
    ap.add_argument("--mode", choices=["train", "predict", "prepare"], default="train",
                    help="train: the contract line (BASELINE config 2).  predict: BASELINE config 3 (informative).  "
                         "prepare: the data-preparation chain in front of the net (informative)")
    return ap.parse_args()


def _leg_in_fresh_process(extra_args, timeout=600):
    """One training-step leg (bf16 mode, dense tiles) in a process of its own: a second net in THIS process shares the
    first one's hardware queues (torch streams are dealt round-robin onto 4 of them) and measured 10 % slower than the
    same leg alone.  Returns the leg's JSON line as a dict."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--skip-cpu-baseline", "--skip-roofline", "--skip-extras"] + extra_args
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    if out.returncode != 0:
        raise RuntimeError(f"leg {extra_args} failed: {out.stderr[-400:]}")
    return json.loads(out.stdout.strip().splitlines()[-1])


def timed(fn, steps, world):
    """Time exactly ``steps`` calls bracketed by barrier + synchronize on both sides; max over ranks (seconds)."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak (= fp32 vector peak)


def _time_launch(launch, reps=20):
    """Average duration (ms) of one launch, HIP events on the launch stream (torch's current stream)."""
    for _ in range(3):
        launch()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        launch()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / reps


def _pmc_traffic(*kernel_prefixes):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/*pmc_fetch_size.csv and
    *pmc_write_size.csv: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of tools/pmc_target.py, per-kernel means in
    KB; since round 2 one row per kernel name AND launch grid, so a level-1 launch is not averaged with the other
    levels').  FETCH_SIZE is doubled: MI355X_MICROARCH.md calibrates it at exactly half the bytes of 16-byte-per-lane
    reads on gfx950, which is what these kernels issue (row gathers and float4 streams); WRITE_SIZE is taken as is.
    ``kernel_prefixes``: the kernel's name in the newest pass first, older names after (kernels were renamed when
    variants were added).  Returns None when no pass in the tree lists the kernel."""
    import csv
    import glob

    fetch = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_fetch_size.csv")), reverse=True)
    write = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_write_size.csv")), reverse=True)

    def lookup(files, col):
        for f in files:                      # newest pass first
            rows = list(csv.DictReader(open(f)))
            for prefix in kernel_prefixes:
                vals = [float(row[col]) for row in rows if row["kernel"].startswith(prefix)]
                if vals:
                    return max(vals) * 1024.0  # several launch geometries of one kernel: the roofline entry is the largest
        return None

    r, w = lookup(fetch, "FETCH_SIZE"), lookup(write, "WRITE_SIZE")
    return None if r is None or w is None else int(2 * r + w)


def stage_rooflines(net, pos, plan):
    """Roofline entries, timed live.  ``dominant``: the kernel with the largest share of the training step —
    lfa_bwd_kernel<64,16,PIPE=true> (block 2 / lfa2: 51 200 centres x 16 neighbours, ch = 64), an fp32-MFMA kernel:
    3 x 2 x n x K x (ch^2 + 10 ch/2) flop per launch (recomputed attention GEMM + its two backward GEMMs).
    ``knn_lse``: the HBM-class kernels of the kNN + LSE-gather stage at level 1 (204 800 points), algorithmic bytes
    per launch from SURVEY 8d: kNN read 12 n + write 4 n K; LFA(ch) read n (12 + 4 ch/2 + 4 K), write 4 n ch."""
    from myria3d_amd import ops

    K = net.num_neighbors
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    with torch.no_grad():
        net.overlap_geometry, keep = False, net.overlap_geometry
        net.batch_geometry, keep_b = False, net.batch_geometry  # per-level launches: the kernels the captured step runs
        geo = net._geometry(pos, plan, None, True)
        net.overlap_geometry, net.batch_geometry = keep, keep_b
        # ---- dominant kernel: LFA backward at level 2, ch = 64
        lfa = net.block2.lfa2
        ch = lfa.mlp_attention.lins[0].weight.shape[0]
        n2, D = geo.pos4[1].shape[0], ch // 2
        xin = torch.randn(n2, D, device=pos.device)
        dout = torch.randn(n2, ch, device=pos.device)
        enc_lin, enc_bn = lfa.mlp_encoder.lins[0], lfa.mlp_encoder.norms[0].module
        wf, bf, _, _ = ops.lfa_enc_fold(enc_lin, enc_bn, None, 0)
        wp, wpt = ops.pack_attention_weights(lfa.mlp_attention.lins[0].weight, True)
        dx = torch.zeros((n2, D), device=pos.device)
        dw = torch.empty((ch, ch), device=pos.device)
        G = torch.empty(11 * D, dtype=torch.float64, device=pos.device)
        ws = torch.empty(ops.lib().m3d_lfa_bwd_workspace_bytes(n2, K, ch), dtype=torch.uint8, device=pos.device)
        ms = _time_launch(lambda: ops.call(
            "m3d_lfa_bwd", xin.data_ptr(), geo.pos4[1].data_ptr(), geo.knn[1].data_ptr(), n2, K, ch, wf.data_ptr(),
            bf.data_ptr(), wp.data_ptr(), wpt.data_ptr(), ops.LRELU_SLOPE, dout.data_ptr(), dx.data_ptr(), dw.data_ptr(),
            0, G.data_ptr(), ws.data_ptr(), st))
        # ALGORITHMIC flops of this backward (backward = 2 x forward: dF = dA W and dW = dA^T F, plus the
        # encoder's two transposes) vs the flops the kernel EXECUTES (it recomputes the forward attention GEMM
        # A = F W^T instead of saving [E, ch] logits: a third GEMM)
        flop_alg = 2 * 2 * n2 * K * (ch * ch + 10 * D)
        flop_exe = 3 * 2 * n2 * K * (ch * ch + 10 * D)
        tf = flop_alg / (ms * 1e-3) / 1e12
        tf_exe = flop_exe / (ms * 1e-3) / 1e12
        out["dominant"] = {"kernel": f"lfa_bwd_kernel<64,16> (block2.lfa2, ch={ch}, n={n2}, K={K}) + partial reduce",
                           "bound": "mfma", "achieved": round(tf, 1), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                           "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
                           "frac_algorithmic": round(tf / FP32_MFMA_PEAK_TF, 4),
                           "frac_executed": round(tf_exe / FP32_MFMA_PEAK_TF, 4),
                           "traffic": _pmc_traffic("void lfa_bwd_kernel<64, 16, true>", "void lfa_bwd_pipe_kernel<64, 16>",
                                                    "void lfa_bwd_kernel<64, 16>"),
                           "algorithmic_flop_per_launch": flop_alg, "executed_flop_per_launch": flop_exe,
                           "avg_launch_ms": round(ms, 4)}
        # ---- kNN + LSE gather stage at level 1
        n1 = geo.pos4[0].shape[0]
        ix = geo.index[0]
        idx = torch.empty((n1, K), dtype=torch.int32, device=pos.device)
        ms_knn = _time_launch(lambda: ops.call(
            "m3d_knn_query", ix.ws.data_ptr(), ix.ptr.data_ptr(), n1, ix.num_clouds, None, 0, ix.ws.data_ptr(),
            ix.ptr.data_ptr(), n1, K, 1, idx.data_ptr(), None, st))
        lfa1 = net.block1.lfa2
        ch1 = lfa1.mlp_attention.lins[0].weight.shape[0]
        x1 = torch.randn(n1, ch1 // 2, device=pos.device)
        wf1, bf1, _, _ = ops.lfa_enc_fold(lfa1.mlp_encoder.lins[0], lfa1.mlp_encoder.norms[0].module, None, 0)
        wp1, _ = ops.pack_attention_weights(lfa1.mlp_attention.lins[0].weight, False)
        o1 = torch.empty((n1, ch1), device=pos.device)
        ms_lfa = _time_launch(lambda: ops.call(
            "m3d_lfa_fwd", x1.data_ptr(), geo.pos4[0].data_ptr(), geo.knn[0].data_ptr(), n1, K, ch1, wf1.data_ptr(),
            bf1.data_ptr(), wp1.data_ptr(), ops.LRELU_SLOPE, o1.data_ptr(), st))
        b_knn = n1 * (12 + 4 * K)
        b_lfa = n1 * (12 + 4 * ch1 // 2 + 4 * K + 4 * ch1)
        out["knn_lse"] = [
            {"kernel": f"knn_query_queue_kernel<16> (level 1, n={n1}, K={K})", "bound": "hbm",
             "achieved": round(b_knn / (ms_knn * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(b_knn / (ms_knn * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": b_knn,
             "traffic": _pmc_traffic("void knn_query_queue_kernel<16, KeyF64", "void knn_query_kernel<16, KeyF64>",
                                     "void knn_query_kernel<16>"),
             "avg_launch_ms": round(ms_knn, 4)},
            {"kernel": f"lfa_fwd_kernel<16,16> (block1.lfa2, ch={ch1}, n={n1}, K={K})", "bound": "hbm",
             "achieved": round(b_lfa / (ms_lfa * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(b_lfa / (ms_lfa * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": b_lfa,
             "traffic": _pmc_traffic("void lfa_fwd_kernel<16, 16>"),
             "avg_launch_ms": round(ms_lfa, 4)}]
    return out


def _pick_threads():
    """Thread count for the CPU baseline: the fastest of {1, 4, 8, 16, all cores} on a micro-probe shaped like the
    oracle's level-1 edge tensors (some hosts — e.g. oversubscribed VMs — are slower with every core)."""
    ncpu = os.cpu_count() or 1
    a, b, w = torch.rand(204800, 16), torch.rand(204800, 16), torch.rand(16, 16)
    best, best_t = 1, float("inf")
    for th in sorted({1, min(4, ncpu), min(8, ncpu), min(16, ncpu), ncpu}):
        torch.set_num_threads(th)
        (a * b) @ w  # warm-up
        t0 = time.perf_counter()
        for _ in range(3):
            torch.exp((a * b) @ w)
        dt = time.perf_counter() - t0
        if dt < best_t * 0.9:
            best, best_t = th, dt
    return best


def cpu_baseline(tiles, points, K, full=False):
    """The CPU oracle (op-for-op restatement of the reference path; kNN through cKDTree like torch_cluster's CPU
    path) on the SAME workload as the GPU line (B = 16 tiles x 12 800 points; fewer tiles only when
    the host has too little free memory), fwd+bwd in train mode (CE loss) and fwd-only in eval mode, median of the
    timed iterations.  Default (bounded to ~30 s of CPU work so the driver's default run stays short): 1 warm-up + 3
    timed iterations at the thread count a micro-probe picks.  ``full`` = BASELINE.md 3 to the letter: 3 warm-up + 10
    timed, at the probe-picked count AND on every host core."""
    import statistics

    from oracle.randla_oracle import RandLANetOracle
    from myria3d_amd.synthetic import synthetic_batch

    ncpu = os.cpu_count() or 1
    try:
        import psutil
        free_gb = psutil.virtual_memory().available / 2**30
    except Exception:
        free_gb = 16.0
    tiles = max(1, min(tiles, int(free_gb // 1.5)))  # ~0.7 GB of autograd intermediates per 12 800-point tile
    picked = _pick_threads()
    torch.manual_seed(0)
    net = RandLANetOracle(9, 6, num_neighbors=K, return_logits=True, knn="kdtree")
    x, pos, batch, ptr, y = synthetic_batch([points] * tiles)
    warm, reps = (3, 10) if full else (1, 3)

    def leg(threads):
        torch.set_num_threads(threads)

        def train_step():
            net.train()
            net.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(net(x, pos, batch, ptr), y).backward()

        def fwd_step():
            net.eval()
            with torch.no_grad():
                net(x, pos, batch, ptr)

        res = {}
        for name, fn in (("fwd_bwd", train_step), ("fwd_only", fwd_step)):
            for _ in range(warm):
                fn()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            res[name] = tiles * points / statistics.median(ts)
        return res

    main = leg(picked)
    out = {"value": round(main["fwd_bwd"], 1), "unit": "points/s", "cores": picked, "kind": "port",
           "fwd_only": round(main["fwd_only"], 1), "host_cores": ncpu,
           "sample": f"{tiles} tiles x {points} pts (the GPU line's batch), median of {reps} timed iteration(s) after "
                     f"{warm} warm-up; fwd+bwd = train mode + CE loss + backward, fwd_only = eval / no_grad; "
                     f"oracle/randla_oracle.py (unfused torch CPU ops, cKDTree kNN); threads = {picked} (fastest of "
                     f"{{1,4,8,16,{ncpu}}} on a micro-probe), host has {ncpu} cores"}
    if full and picked != ncpu:
        allc = leg(ncpu)
        out["all_cores"] = {"cores": ncpu, "value": round(allc["fwd_bwd"], 1), "fwd_only": round(allc["fwd_only"], 1)}
    return out


def predict_bench(args, dev, world=1, rank=0, reps=None):
    """BASELINE config 3 (informative, not the contract line): predict.py-shaped inference over a synthetic 1 km^2
    cloud = 400 tiles of 50 m, batches of 50 tiles (configs/experiment/predict.yaml:21-23).  Per batch: eval forward
    on the sub-sampled tiles (12 800 points each), knn_interpolate(k=10) of the logits to every point of the full
    tiles (25 000 each; the reference does this on the CPU, model.py:86-103), scatter_sum into the per-cloud logit
    accumulator by original index (interpolation.py:116); finally softmax / argmax / entropy over all points.
    N > 1: the 8 batches shard over the ranks (independent units: NO collective on the data path; every rank merges
    and classifies its own tiles, as when each rank writes its own LAS files)."""
    from myria3d_amd import HipRandLANet, knn_interpolate, make_plan, predict_reduce, scatter_sum
    from myria3d_amd.ddp import shard_tiles
    from myria3d_amd.synthetic import synthetic_tile

    n_full, n_sub, tiles, bs, C = 25000, 12800, 400, 50, 7
    g = torch.Generator().manual_seed(1)
    full_pos, sub_sel, feats = [], [], []
    for tid in range(bs):  # 50 distinct tiles, re-used for the 8 batches of the sweep
        x, pos, _ = synthetic_tile(n_full, tid, 9, C)
        sel = torch.randperm(n_full, generator=g)[:n_sub].sort().values
        full_pos.append(pos), sub_sel.append(sel), feats.append(x)
    pos_full = torch.cat(full_pos).to(dev)
    x_full = torch.cat(feats).to(dev)
    sel = torch.cat([s_ + i * n_full for i, s_ in enumerate(sub_sel)]).to(dev)
    pos_sub, x_sub = pos_full[sel].contiguous(), x_full[sel].contiguous()
    ptr_sub = torch.arange(0, (bs + 1) * n_sub, n_sub, dtype=torch.int64, device=dev)
    batch_sub = torch.arange(bs, device=dev).repeat_interleave(n_sub)
    batch_full = torch.arange(bs, device=dev).repeat_interleave(n_full)
    torch.manual_seed(0)
    net = HipRandLANet(9, C, num_neighbors=16, return_logits=True).to(dev).eval()
    plan = make_plan(ptr_sub.tolist(), 4, 16, dev)
    mine = shard_tiles(tiles // bs, rank, world)  # batches of this rank
    total = tiles * n_full
    acc = torch.zeros((max(1, len(mine)) * bs * n_full, C), device=dev)
    orig = torch.arange(bs * n_full, dtype=torch.int32, device=dev)

    def sweep():
        acc.zero_()
        with torch.no_grad():
            for b in range(len(mine)):
                logits = net(x_sub, pos_sub, None, ptr_sub, plan=plan)
                dense = knn_interpolate(logits, pos_sub, pos_full, batch_sub, batch_full, k=10)
                scatter_sum(dense, orig + b * bs * n_full, out=acc)
            probas, pred, entropy = predict_reduce(acc)  # softmax / argmax / entropy in one launch (interp.hip)
        return pred, entropy

    for _ in range(max(1, args.warmup // 3)):
        sweep()
    reps = reps if reps is not None else max(1, args.steps // 10)
    dt = timed(sweep, reps, world) / reps
    return {"metric": "points/sec classified, predict path (fwd + kNN-interpolation k=10 + merge)",
            "value": round(total / dt, 1), "unit": "points/s", "n_gpus": world, "ms_per_sweep": round(dt * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 3: {tiles} tiles x {n_full} pts (sub-sampled to {n_sub}), "
                                   f"batch {bs}, K=16, C={C}, interpolation k=10; batches sharded over {world} rank(s), "
                                   "no data-path collective",
                       "launch": "eager"}}


def prepare_bench(args, dev):
    """Data preparation in front of the net (informative, not the contract line): raw tiles of
    ~80 000 points (a 50 m Lidar-HD tile at ~30 pts/m^2) through GridSampling(0.25) -> MinimumNumNodes(300) ->
    MaximumNumNodes(40000) -> Center -> NullifyLowestZ -> NormalizePos -> StandardizeRGBAndIntensity
    (configs/datamodule/transforms/preparations/points_budget.yaml, normalizations/default.yaml) as ONE batch on the
    device, next to the oracle's per-tile CPU chain (oracle/prep_oracle.py) on a bounded sample of the same tiles."""
    import numpy as np

    from myria3d_amd import transforms as T

    n_raw, B = 80000, args.tiles
    rs = np.random.RandomState(0)
    pos = torch.from_numpy((rs.uniform(0, 1, (B * n_raw, 3)) * np.array([50.0, 50.0, 6.0])).astype(np.float32))
    pos += torch.arange(B).repeat_interleave(n_raw)[:, None].float() * torch.tensor([50.0, 0.0, 0.0])
    x = torch.from_numpy(rs.uniform(0, 1, (B * n_raw, 9)).astype(np.float32))
    x[:, 0] = torch.from_numpy(rs.gamma(2.0, 300.0, B * n_raw).astype(np.float32))
    x[:, 7] = torch.from_numpy(rs.uniform(0, 255, B * n_raw).astype(np.float32))
    y = torch.from_numpy(rs.randint(0, 6, B * n_raw).astype(np.int64))
    ptr = torch.arange(0, (B + 1) * n_raw, n_raw, dtype=torch.int64)
    pos_d, x_d, y_d, ptr_d = pos.to(dev), x.to(dev), y.to(dev), ptr.to(dev)

    def chain():
        p, xx, yy, pt = T.grid_sampling(pos_d, x_d, y_d, ptr_d, 0.25)
        p, xx, yy, pt, _ = T.node_budget(p, xx, yy, pt, minimum=300, maximum=40000, seed=1)
        p, xx = T.normalize_tiles(p, xx, pt, center=True, nullify_z=True, subtile_width=50, intensity_col=0, rgb_col=7)
        return p, xx, yy, pt

    for _ in range(max(1, args.warmup // 2)):
        out = chain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = max(1, args.steps // 2)
    for _ in range(reps):
        out = chain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res = {"metric": "raw points/sec through the data preparation (GridSampling + node budget + normalisations)",
           "value": round(B * n_raw / dt, 1), "unit": "points/s", "n_gpus": 1, "ms_per_batch": round(dt * 1e3, 3),
           "higher_is_better": True, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{B} raw tiles x {n_raw} pts, voxel 0.25 m -> {int(out[3][-1])} points kept",
                      "launch": "eager (one host sync per batch: the voxel count)"}}
    if not args.skip_cpu_baseline:
        sys.path.insert(0, ROOT)
        from oracle import prep_oracle as O  # checker / baseline only

        k = min(args.cpu_tiles, B)
        torch.set_num_threads(1)  # the reference runs these transforms in single-threaded dataloader workers
        t0 = time.perf_counter()
        O.prepare_tiles(pos[:k * n_raw], x[:k * n_raw], y[:k * n_raw], ptr[:k + 1].tolist(), 0.25, 50, 0, 7)
        dt_c = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(k * n_raw / dt_c, 1), "unit": "points/s", "cores": 1, "kind": "port",
                               "sample": f"{k} tile(s) x {n_raw} raw pts, oracle/prep_oracle.py (torch CPU ops, one "
                                         "thread = one dataloader worker; the reference uses 3 workers)"}
    print(json.dumps(res), flush=True)


def _baseline_config(points: int, neighbors: int) -> str:
    if (points, neighbors) == (12800, 16):
        return "BASELINE config 2"
    if (points, neighbors) == (40000, 32):
        return "BASELINE config 5, RandLA part"
    return "non-BASELINE size"


def _respawn(args):
    """``python bench.py --gpus N`` without a launcher: re-execute under torch.distributed.run with N ranks."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def _dry_run_gloo(args, world, rank):
    """Launch check without GPUs: the N ranks rendezvous over gloo, one all-reduce, rank 0 prints the census."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    t = torch.ones(1)
    dist.all_reduce(t)
    pids = [None] * world
    dist.all_gather_object(pids, os.getpid())
    if rank == 0:
        print(json.dumps({"dry_run": True, "backend": "gloo", "n_gpus": world, "ranks": int(t.item()), "pids": pids}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def train_bench(args, dev, world, rank, B, N, K, steps, warmup, with_eager=False, precision="fp32"):
    """The contract measurement: ``steps`` training steps (fwd + CE + bwd + [all-reduce] + Adam) and ``steps`` eval
    forwards over B tiles x N points per rank.  Returns (record, net, pos, plan)."""
    from myria3d_amd import FusedAdam, HipRandLANet, cross_entropy, make_plan
    from myria3d_amd.ddp import broadcast_module_state, shard_tiles
    from myria3d_amd.synthetic import synthetic_batch

    tile_ids = shard_tiles(B * world, rank, world)  # weak scaling: B tiles per rank
    x, pos, batch, ptr, y = synthetic_batch([N] * B, first_tile_id=tile_ids.start)
    x, pos, ptr, y = x.to(dev), pos.to(dev), ptr.to(dev), y.to(dev)
    torch.manual_seed(0)
    net = HipRandLANet(9, 6, decimation=4, num_neighbors=K, return_logits=True).to(dev)
    net.matmul_precision = precision
    # every parameter / gradient becomes a view of one flat buffer: the backward kernels accumulate into it, RCCL
    # all-reduces it as ONE 4.45 MB bucket, m3d_adam_step updates (and clears) it in one launch
    net.flatten_parameters()
    broadcast_module_state(net)
    plan = make_plan(ptr.tolist(), 4, K, dev)
    opt = FusedAdam(net, lr=0.003933709606504788, all_reduce=True)  # lr: configs/model/pyg_randla_net_model.yaml:4

    look = args.lookahead

    def fwd_bwd(prefetch=look):
        if not net.training:
            net.train()  # (walks 266 modules: ~0.3 ms of host time when every step pays it)
        if prefetch:  # the NEXT step's kNN tables / decimation, enqueued stage by stage BETWEEN the blocks of this forward
            net.prefetch_geometry(pos, ptr, plan, train=True, interleave=True)
        out = net(x, pos, None, ptr, plan=plan)  # (lookahead: consumes the tables the previous step prefetched)
        loss = cross_entropy(out, y, ignore_index=65)  # configs/model/criterion/CrossEntropyLoss.yaml
        loss.backward()
        if net.grad_side is not None:
            net.grad_side.join()  # weight-gradient side stream rejoins (must happen inside a captured region)
        if look:
            net.join_geometry()

    def train_step(prefetch=look):
        fwd_bwd(prefetch)
        opt.step()  # (N>1: ONE flat-gradient all-reduce over RCCL) + Adam + gradient clear

    def fwd_step(prefetch=look):
        if net.training:
            net.eval()
        with torch.no_grad():
            if prefetch:
                net.prefetch_geometry(pos, ptr, plan, train=False, interleave=True)
            net(x, pos, None, ptr, plan=plan)
            if look:
                net.join_geometry()

    def geo_step(train):
        """The position-only work of the NEXT step as a stand-alone unit (its own hipGraph in the dual-graph launch)."""
        net.prefetch_geometry(pos, ptr, plan, train=train)
        net.join_geometry()

    def capture_dual(body, train):
        """Two hipGraphs per buffer set instead of one: B = the step (consumes the tables prefetched one step earlier), A =
        the position-only work for the step after it.  Replayed on two streams they run CONCURRENTLY — inside one graph
        the executor submits the position-only branch first and the feature chain starts ~0.8 ms into every replay
        (profiles/r02u_step_timeline.csv).  Ordering between the streams: B_i waits for A_{i-1} (its tables), A_i waits
        for B_{i-1} (the last reader of the buffer set A_i rewrites)."""
        gB, gA = [], []
        for _ in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                body()
            gB.append(g)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                geo_step(train)
            gA.append(g)
        sA = torch.cuda.Stream()
        evA, evB = torch.cuda.Event(), torch.cuda.Event()
        evA.record()
        evB.record()
        turn = [0]

        def step():
            k = turn[0] & 1
            cur = torch.cuda.current_stream()
            cur.wait_event(evA)
            gB[k].replay()
            sA.wait_event(evB)
            with torch.cuda.stream(sA):
                gA[k].replay()
                evA.record(sA)
            evB.record(cur)
            turn[0] += 1

        return step

    launch = "eager"
    step_fn, fwd_fn = train_step, fwd_step
    eager_ms = None
    mode = "eager" if args.no_graph else args.launch
    probe = {}
    if with_eager or mode == "auto":  # what a Lightning loop (no capture, model.py:79) sees: the same kernels, launched one by one
        for _ in range(40):  # (the eager path needs ~30 steps to settle: 6.2 -> 5.55 ms, tools/scratch/eager_ramp.py)
            train_step()
        eager_ms = probe["eager"] = timed(train_step, 15, world) / 15 * 1e3
    # hipGraph: the forward+loss+backward launch sequence (parallel branches for the position-only work and the weight
    # gradients) is captured once and replayed.  With N > 1 the optimizer (all-reduce + 2 launches) stays outside the
    # graph so that no collective is captured
    if mode != "eager":
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    train_step()
                    fwd_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if look:  # the captured steps must CONSUME tables prefetched one step earlier: leave one pending.  Two steps:
                # the warm-up above alternated train and eval prefetches, so one of the two buffer sets still has the
                # eval layout — a captured prefetch into it would re-create it (25 clone copies per replay)
                train_step()
                train_step()
                torch.cuda.synchronize()
            # with the lookahead the prefetched tables live in two buffer sets used in turn: one captured step per set
            if look and args.lookahead_mode == "dual":
                dual = capture_dual((lambda: train_step(False)) if world == 1 else (lambda: fwd_bwd(False)), True)

                def graph_step():
                    dual()
                    if world > 1:
                        opt.step()  # RCCL all-reduce + Adam stay outside the captured graph
            else:
                g_train = []
                for _ in range(2 if look else 1):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):  # (RCCL's watchdog thread must not void the capture)
                        if world == 1:
                            train_step()  # no collective: the optimizer launches are part of the graph
                        else:
                            fwd_bwd()
                    g_train.append(g)
                turn = [0]

                def graph_step():
                    g_train[turn[0] % len(g_train)].replay()
                    turn[0] += 1
                    if world > 1:
                        opt.step()  # RCCL all-reduce + Adam stay outside the captured graph

            step_fn, launch = graph_step, "hipgraph"
        except Exception as e:  # capture is an optimisation, never a requirement
            if rank == 0:
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            step_fn, launch = train_step, "eager"

    if mode == "auto" and launch == "hipgraph":
        for _ in range(5):
            step_fn()
        probe["hipgraph"] = timed(step_fn, 10, world) / 10 * 1e3
        if probe["eager"] < probe["hipgraph"]:
            step_fn, launch = train_step, "eager"
    for _ in range(warmup):
        step_fn()
    dt = timed(step_fn, steps, world)
    # eval forward of the trained weights.  The first (eager) pass folds the BatchNorms / packs the attention weights
    # (cached by the module until the next training phase); the captured graph then holds the per-batch work only
    fwd_step()
    fwd_step()  # (lookahead: the second call consumes what the first one prefetched and leaves an eval-mode slot pending)
    fprobe = {}
    if mode == "auto":
        for _ in range(20):
            fwd_step()
        fprobe["eager"] = timed(fwd_step, 15, world) / 15 * 1e3
    if mode != "eager":
        try:
            torch.cuda.synchronize()
            if look and args.lookahead_mode == "dual":
                fwd_graph = capture_dual(lambda: fwd_step(False), False)
            else:
                g_fwd = []
                for _ in range(2 if look else 1):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        fwd_step()
                    g_fwd.append(g)
                fturn = [0]

                def fwd_graph():
                    g_fwd[fturn[0] % len(g_fwd)].replay()
                    fturn[0] += 1

            fwd_fn = fwd_graph
        except Exception as e:
            if rank == 0:
                print(f"[bench] eval hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
    if mode == "auto" and fwd_fn is not fwd_step:
        for _ in range(3):
            fwd_fn()
        fprobe["hipgraph"] = timed(fwd_fn, 10, world) / 10 * 1e3
        if fprobe["eager"] < fprobe["hipgraph"]:
            fwd_fn = fwd_step
    for _ in range(max(1, warmup // 2)):
        fwd_fn()
    dt_f = timed(fwd_fn, steps, world)

    total_points = B * N * world
    res = {
        "metric": f"points/sec fwd+bwd, RandLA-Net, {N // 1000} {N % 1000:03d}-pt tiles",
        "value": round(total_points * steps / dt, 1),
        "unit": "points/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if precision == "fp32" else "f32 storage / accumulate, bf16 matrix-core operands",
        "data": "synthetic",
        "config": {"workload": f"RandLA-Net train step (fwd + CE + bwd + grad all-reduce + Adam), {B} tiles x {N} pts per GPU, "
                               f"K={K}, F=9, C=6, decimation 4 ({_baseline_config(N, K)}, {precision})",
                   "tiles_per_gpu": B, "points_per_tile": N, "num_neighbors": K, "parallelism": f"dp{world} over tiles",
                   "collective": "one flat 4.45 MB fp32 gradient all-reduce per step (RCCL)" if world > 1 else "none (1 rank)",
                   "launch": launch, **({"geometry_lookahead": args.lookahead_mode} if look else {})},
        "fwd_only": {"value": round(total_points * steps / dt_f, 1), "unit": "points/s",
                     "ms_per_step": round(dt_f / steps * 1e3, 4), "mode": "eval, no_grad",
                     "launch": "eager" if fwd_fn is fwd_step else "hipgraph"},
    }
    if probe:
        res["launch_probe_ms"] = {k: round(v, 4) for k, v in probe.items()}
    if eager_ms is not None:
        res["eager_ms_per_step"] = round(eager_ms, 4)
    return res, net, pos, plan


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _respawn(args)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or drop the launcher: `python bench.py --gpus N` starts its own ranks)")
    if args.dry_run_gloo:
        _dry_run_gloo(args, world, rank)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) are visible")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" IS RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: RCCL came up with {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from myria3d_amd import _lib

    _lib.lib()  # no fallback: fail here if the HIP library is missing
    if args.mode == "predict":
        res = predict_bench(args, dev, world, rank)
        if rank == 0:
            print(json.dumps(res), flush=True)
    elif args.mode == "prepare":
        if world > 1:
            raise SystemExit("--mode prepare is a single-GPU run")
        prepare_bench(args, dev)
    else:
        B, N, K = args.tiles, args.points, args.neighbors
        extras = world == 1 and not args.skip_extras
        res, net, pos, plan = train_bench(args, dev, world, rank, B, N, K, args.steps, args.warmup, with_eager=extras,
                                          precision=args.precision)
        if world > 1:
            res["rccl_ranks"] = dist.get_world_size()
        if rank == 0:
            if not args.skip_roofline:
                try:
                    rl = stage_rooflines(net, pos, plan)
                    res["roofline"] = rl["dominant"]
                    res["roofline_knn_lse_stage"] = rl["knn_lse"]
                except Exception as e:
                    res["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        del net, pos, plan
        if extras:
            # informative extra legs of the N=1 line (BASELINE configs 3 and 5), short: they share the driver's clock
            try:
                torch.cuda.empty_cache()
                pr = predict_bench(args, dev, reps=2)
                res["predict_config3"] = {k: pr[k] for k in ("value", "unit", "ms_per_sweep")} | {"workload": pr["config"]["workload"]}
            except Exception as e:
                res["predict_config3"] = {"error": f"{type(e).__name__}: {e}"}
            try:  # BASELINE config 2 names bf16: the same step with the matrix-bound layers on bf16 matrix cores (fp32
                # accumulate; parity bar of SURVEY 8c: logits within 3e-2 of the fp32 oracle, tests/test_gpu_net.py)
                torch.cuda.empty_cache()
                b16 = _leg_in_fresh_process(["--precision", "bf16", "--steps", str(args.steps), "--warmup", str(args.warmup),
                                             "--tiles", str(B), "--points", str(N), "--neighbors", str(K)])
                res["bf16"] = {"value": b16["value"], "unit": "points/s", "ms_per_step": b16["ms_per_step"],
                               "fwd_only": b16["fwd_only"],
                               "what": "LFA attention GEMMs (ch >= 64, fwd + bwd) and SharedMLP GEMMs with K > 64 (fwd + "
                                       "dgrad; deep-layer wgrad) on v_mfma_f32_16x16x32_bf16, fp32 accumulate; storage / kNN / "
                                       "softmax / BatchNorm / level-1 GEMMs fp32"}
            except Exception as e:
                res["bf16"] = {"error": f"{type(e).__name__}: {e}"}
            if (N, K) == (12800, 16):
                try:
                    torch.cuda.empty_cache()
                    d5 = _leg_in_fresh_process(["--steps", "5", "--warmup", "2", "--tiles", "16", "--points", "40000",
                                                "--neighbors", "32"])
                    res["dense_tiles_config5"] = {"value": d5["value"], "unit": "points/s", "ms_per_step": d5["ms_per_step"],
                                                  "fwd_only": d5["fwd_only"], "workload": d5["config"]["workload"]}
                except Exception as e:
                    res["dense_tiles_config5"] = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            if world == 1 and not args.skip_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args.cpu_tiles, N, K, full=args.cpu_baseline_full)
            print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
