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
_T0 = time.perf_counter()


def _progress(msg):
    """Leg-by-leg progress on stderr (the driver's clock runs around the whole command: a slow leg must be findable)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps (default 200: a timed region of ~0.8 s; rounds 1-4 defaulted to 20 = 86 ms, within the "
                         "box-to-box noise)")
    ap.add_argument("--warmup", type=int, default=10)
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
    ap.add_argument("--precision", choices=("fp32", "bf16", "bf16ops", "bf16x3"), default="fp32",
                    help="precision mode of the timed net (the contract line is fp32; 'bf16' = bf16 activation storage + bf16 "
                         "matrix-core operands: the \"bf16\" leg; 'bf16ops' = the operands alone, fp32 storage: rounds 2-5's leg)")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-roofline", action="store_true")
    ap.add_argument("--skip-extras", action="store_true",
                    help="skip the informative extra legs of the N=1 line (eager step, predict sweep, dense tiles)")
    ap.add_argument("--skip-legs", default="", help="comma list of informative legs to leave out of the N=1 line: "
                    "predict,bf16,bf16x3,dropin,collective,torch,dense,pointnet2")
    ap.add_argument("--cpu-tiles", type=int, default=16, help="tiles in the CPU-baseline sample (BASELINE.md 3: 16)")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="also time the CPU oracle on ALL host cores (3 + 10 iterations more; several minutes of CPU time)")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="launch check without GPUs: bring the N ranks up over gloo, exchange one all-reduce, print the "
                         "rank census (used by the CPU tests)")
    ap.add_argument("--mode", choices=["train", "predict", "prepare", "dropin", "pointnet2"], default="train",
                    help="train: the contract line (BASELINE config 2).  predict: BASELINE config 3 (informative).  "
                         "prepare: the data-preparation chain in front of the net (informative).  dropin: the plain "
                         "Lightning-style step (no plan / prefetch / graph / flat buffers, torch loss and Adam)")
    ap.add_argument("--collective", choices=("captured", "eager"), default="captured",
                    help="with a gradient exchange: the RCCL all-reduce + Adam inside the step's hipGraph (default) or after it")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: bring up a 1-rank RCCL process group and take the N > 1 code path (captured fwd + bwd, "
                         "out-of-graph all-reduce of the flat gradient bucket, eager Adam) — exercises the collective and "
                         "its interplay with hipGraph capture on a one-GPU box")
    return ap.parse_args()


def _leg_in_fresh_process(extra_args, timeout=240):
    """One training-step leg (bf16 mode, dense tiles) in a process of its own: a second net in THIS process shares the
    first one's hardware queues (torch streams are dealt round-robin onto 4 of them) and measured 10 % slower than the
    same leg alone.  Returns the leg's JSON line as a dict."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--skip-cpu-baseline", "--skip-roofline", "--skip-extras"] + extra_args
    _progress("leg in a process of its own: " + " ".join(extra_args))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    if out.returncode != 0:
        raise RuntimeError(f"leg {extra_args} failed: {out.stderr[-400:]}")
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]  # (RCCL prints its version banner on stdout)
    if not lines:
        raise RuntimeError(f"leg {extra_args} printed no JSON line: {out.stdout[-300:]}")
    return json.loads(lines[-1])


LAST_RANK_SECONDS = None  # per-rank seconds of the most recent timed() region BEFORE its closing barrier (N > 1 only)


def timed(fn, steps, world):
    """Time exactly ``steps`` calls bracketed by barrier + synchronize on both sides; max over ranks (seconds).  With
    N > 1 every rank's own time up to its synchronize (i.e. before the closing barrier makes them equal) is gathered into
    ``LAST_RANK_SECONDS``, so that the line can show a slow rank."""
    global LAST_RANK_SECONDS
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([own], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        LAST_RANK_SECONDS = [float(v.item()) for v in every]
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


class PmcMissing(RuntimeError):
    pass


def _pmc_traffic(prefix, grid=None):
    """HBM-side bytes per launch of a kernel from the NEWEST committed rocprofv3 PMC pass (profiles/*pmc_fetch_size.csv
    and *pmc_write_size.csv: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of tools/pmc_target.py, per-kernel means
    in KB, one row per kernel name AND launch grid).  FETCH_SIZE is doubled: MI355X_MICROARCH.md calibrates it at
    exactly half the bytes of 16-byte-per-lane reads on gfx950, which is what these kernels issue (row gathers and
    float4 streams); WRITE_SIZE is taken as is.  ``prefix``: start of the kernel's name in that pass (template
    arguments that were appended since are covered by the prefix match); ``grid``: pick the row of that launch grid
    (several levels run the same template).  Raises PmcMissing when the newest pass does not list the kernel — an older
    round's row for a kernel that has changed since is not evidence."""
    import csv
    import glob

    def newest(pattern):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        if not files:
            raise PmcMissing(f"no profiles/{pattern}")
        return files[-1]

    def lookup(path, col):
        rows = [r for r in csv.DictReader(open(path)) if r["kernel"].startswith(prefix)]
        if grid is not None:
            grids = grid if isinstance(grid, (tuple, list)) else (grid,)
            rows = [r for r in rows if any(r["kernel"].rstrip().endswith(f"grid={g}") for g in grids)] or rows
        if not rows:
            raise PmcMissing(f"{os.path.basename(path)} has no row for {prefix!r}")
        return max(float(r[col]) for r in rows) * 1024.0

    r = lookup(newest("*pmc_fetch_size.csv"), "FETCH_SIZE")
    w = lookup(newest("*pmc_write_size.csv"), "WRITE_SIZE")
    return int(2 * r + w)


VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 4  # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz (MI355X_MICROARCH.md)


def _pmc_sq(prefix, grid=None):
    """``{counter: mean per launch}`` of a kernel from the NEWEST committed SQ pass (profiles/*pmc_sq_roofline_kernels.csv,
    tools/gpu_pmc.sh over tools/pmc_target.py), or None."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_sq_roofline_kernels.csv")))
    if not files:
        return None
    rows = [r for r in csv.DictReader(open(files[-1])) if r["kernel"].startswith(prefix)]
    if grid is not None:
        grids = grid if isinstance(grid, (tuple, list)) else (grid,)
        rows = [r for r in rows if any(r["kernel"].rstrip().endswith(f"grid={g}") for g in grids)] or rows
    if not rows:
        return None
    r = max(rows, key=lambda r: float(r.get("SQ_INSTS_VALU", 0) or 0))
    return {k: float(v) for k, v in r.items() if k not in ("kernel",) and v not in ("", None)} | {"file": os.path.basename(files[-1])}


def _valu_fields(prefix, units, ms, min_lane_ops_per_unit, unit_name, grid=None):
    """The roofline that actually binds the kNN / level-1 LSE kernels: VALU issue.  ``valu_issue_frac`` = the
    wave-level VALU instructions the newest SQ pass counted for this kernel x 4 cycles / (1 024 SIMDs x launch time x 2.4 GHz):
    how busy the VALU issue ports are; ``valu_algorithmic_frac`` = the stated MINIMUM of lane operations per unit (an FMA = 1)
    x units / 64 lanes against the same peak: how far the kernel is from the least VALU work the arithmetic needs."""
    sq = _pmc_sq(prefix, grid)
    out = {"valu_peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS, f"min_lane_ops_per_{unit_name}": min_lane_ops_per_unit,
           "valu_algorithmic_frac": round(min_lane_ops_per_unit * units / 64.0 / (ms * 1e-3) / VALU_PEAK_WAVE_INSTS, 4)}
    if sq and sq.get("SQ_INSTS_VALU"):
        out["valu_insts_per_launch"] = sq["SQ_INSTS_VALU"]
        out[f"valu_insts_per_64_{unit_name}s"] = round(sq["SQ_INSTS_VALU"] / (units / 64.0), 1)
        out["valu_issue_frac"] = round(sq["SQ_INSTS_VALU"] / (ms * 1e-3) / VALU_PEAK_WAVE_INSTS, 4)
        out["sq_pass"] = sq["file"]
    else:
        out["valu_insts_per_launch"] = None
    return {"valu": out}


def _traffic_fields(prefix, grid=None):
    try:
        return {"traffic": _pmc_traffic(prefix, grid)}
    except PmcMissing as e:
        return {"traffic": None, "traffic_error": str(e)}


def stage_rooflines(net, pos, plan):
    """Roofline entries, timed live with HIP events on the launch stream.  ``dominant``: the kernel with the largest
    share of the training step — lfa_bwd_kernel<64,16> (block 2 / lfa2: 51 200 centres x 16 neighbours, ch = 64), an
    fp32-MFMA kernel: 3 x 2 x n x K x (ch^2 + 10 ch/2) flop per launch (recomputed attention GEMM + its two backward
    GEMMs).  ``knn_lse``: the HBM-class kernels of the kNN + LSE-gather stage with the algorithmic bytes of SURVEY 8d:
    kNN read 12 n + write 4 n K (all four levels); LFA(ch) forward read n (12 + 4 ch/2 + 4 K), write 4 n ch; LFA(ch)
    backward read n (12 + 4 ch/2 + 4 ch + 4 K), write 4 n ch/2 — both layers of level 1 (ch 8 and 16)."""
    from myria3d_amd import ops

    K = net.num_neighbors
    st = torch.cuda.current_stream().cuda_stream
    dev = pos.device
    out = {}

    def lfa_operands(lfa, lvl, geo):
        ch = lfa.mlp_attention.lins[0].weight.shape[0]
        n, D = geo.pos4[lvl].shape[0], ch // 2
        enc_lin, enc_bn = lfa.mlp_encoder.lins[0], lfa.mlp_encoder.norms[0].module
        wf, bf, _, _ = ops.lfa_enc_fold(enc_lin, enc_bn, None, 0)
        wp, wpt = ops.pack_attention_weights(lfa.mlp_attention.lins[0].weight, True)
        return ch, n, D, wf, bf, wp, wpt

    def time_lfa_fwd(lfa, lvl, geo):
        ch, n, D, wf, bf, wp, _ = lfa_operands(lfa, lvl, geo)
        x, o = torch.randn(n, D, device=dev), torch.empty((n, ch), device=dev)
        ms = _time_launch(lambda: ops.call(
            "m3d_lfa_fwd", x.data_ptr(), geo.pos4[lvl].data_ptr(), geo.knn[lvl].data_ptr(), n, K, ch, wf.data_ptr(),
            bf.data_ptr(), wp.data_ptr(), ops.LRELU_SLOPE, o.data_ptr(), ops.LFA_FULL, st))
        return ch, n, ms

    def time_lfa_bwd(lfa, lvl, geo):
        """(ms of the kernel as the training step launches it — flags 1 | 2 | 4 | 8: dW_att accumulated, G pre-zeroed, the
        partial-sum reduce deferred to the end of the backward pass, complete neighbourhoods: ONE kernel, what the
        rocprofv3 kernel trace lists under its name —, ms of the self-contained call with its memsets + reduce)"""
        ch, n, D, wf, bf, wp, wpt = lfa_operands(lfa, lvl, geo)
        xin, dout = torch.randn(n, D, device=dev), torch.randn(n, ch, device=dev)
        dx, dw = torch.zeros((n, D), device=dev), torch.zeros((ch, ch), device=dev)
        G = torch.zeros(11 * D, dtype=torch.float64, device=dev)
        ws = torch.empty(ops.lib().m3d_lfa_bwd_workspace_bytes(n, K, ch), dtype=torch.uint8, device=dev)

        def launch(flags):
            return lambda: ops.call(
                "m3d_lfa_bwd", xin.data_ptr(), geo.pos4[lvl].data_ptr(), geo.knn[lvl].data_ptr(), n, K, ch, wf.data_ptr(),
                bf.data_ptr(), wp.data_ptr(), wpt.data_ptr(), ops.LRELU_SLOPE, dout.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                flags, G.data_ptr(), ws.data_ptr(), st)
        return ch, n, D, _time_launch(launch(1 | 2 | 4 | 8)), _time_launch(launch(8))

    def time_lfa_bwd_edge_rows(lfa, lvl, geo):
        """The 8 / 16-channel layers as the training step runs them since round 5: the wave-autonomous kernel storing its
        input gradient per edge (flags | 32) + the sum over every point's reverse neighbour list.  (ms of both, ms of the
        kernel alone)"""
        ch, n, D, wf, bf, wp, wpt = lfa_operands(lfa, lvl, geo)
        xin, dout = torch.randn(n, D, device=dev), torch.randn(n, ch, device=dev)
        dxe, dw = torch.empty((n * K, D), device=dev), torch.zeros((ch, ch), device=dev)
        dx = torch.empty((n, D), device=dev)
        G = torch.zeros(11 * D, dtype=torch.float64, device=dev)
        ws = torch.empty(ops.lib().m3d_lfa_bwd_workspace_bytes(n, K, ch), dtype=torch.uint8, device=dev)
        rptr, rinv, rslot = ops.knn_reverse(geo.knn[lvl])
        slots = ops.USE_LFA_EDGE_SLOTS

        def kernel():
            ops.call("m3d_lfa_bwd_edge_rows", xin.data_ptr(), geo.pos4[lvl].data_ptr(), geo.knn[lvl].data_ptr(), n, K, ch,
                     wf.data_ptr(), bf.data_ptr(), wp.data_ptr(), wpt.data_ptr(), ops.LRELU_SLOPE, dout.data_ptr(), dxe.data_ptr(),
                     rslot.data_ptr() if slots else None, dw.data_ptr(), 1 | 2 | 4, G.data_ptr(), ws.data_ptr(), st)

        def both():
            kernel()
            ops.call("m3d_gather_sum_rows", dxe.data_ptr(), D, rptr.data_ptr(), None if slots else rinv.data_ptr(), dx.data_ptr(),
                     D, n, D, 2, st)
        return ch, n, D, _time_launch(both), _time_launch(kernel)

    def hbm_entry(kernel, nbytes, ms, prefix, grid=None):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": nbytes,
                **_traffic_fields(prefix, grid), "avg_launch_ms": round(ms, 4)}

    # ---- the dominant kernel INSIDE training steps: HIP events around block2.lfa2's backward launch (the launch stream) over
    # eagerly launched fwd + CE + bwd steps on the bench's batch — the same kernel, operands and neighbourhood of launches as
    # in the timed region (whose hipGraph replay cannot carry events around one node)
    in_step = None
    try:
        from myria3d_amd import cross_entropy

        n2_, was_training = plan.totals[1], net.training
        xs = torch.randn(pos.shape[0], net.fc0.weight.shape[1], device=dev)
        ys = torch.randint(0, net.fc_classif.weight.shape[0], (pos.shape[0],), device=dev)
        # (round 6: the three big launches of the family — ch 64 / 128 / 256, the lfa2 layers of blocks 2-4 — are bracketed)
        big_keys = {(plan.totals[1], 64): net.block2.lfa2, (plan.totals[2], 128): net.block3.lfa2, (plan.totals[3], 256): net.block4.lfa2}
        ops.LFA_BWD_TIMER = {"key": (n2_, 64), "keys": set(big_keys), "events": []}
        net.train()
        for _ in range(12):
            cross_entropy(net(xs, pos, None, plan.ptrs[0], plan=plan), ys, 65).backward()
            if net.grad_side is not None:
                net.grad_side.join()
        torch.cuda.synchronize()
        by_key = ops.LFA_BWD_TIMER.get("by_key", {})
        evs = by_key.get((n2_, 64), [])[4:]  # (the first steps settle the allocator / arena)
        if evs:
            in_step = {"ms": sum(a.elapsed_time(b) for a, b in evs) / len(evs), "launches": len(evs),
                       "flags": ops.LFA_BWD_TIMER.get("flags")}
            in_step["family"] = {}
            for (nn_, cc_), ev_ in by_key.items():
                ev_ = ev_[4:]
                if ev_:
                    in_step["family"][cc_] = {"n": nn_, "ms": sum(a.elapsed_time(b) for a, b in ev_) / len(ev_)}
        if net.flat_grads is not None:
            net.flat_grads.zero_()
        net.train(was_training)
    except Exception as e:  # the isolated launch below still gives the entry
        in_step = {"error": f"{type(e).__name__}: {e}"}
    finally:
        ops.LFA_BWD_TIMER = None
    with torch.no_grad():
        net.overlap_geometry, keep = False, net.overlap_geometry
        net.batch_geometry, keep_b = False, net.batch_geometry  # per-level launches: the kernels the captured step runs
        geo = net._geometry(pos, plan, None, True)
        net.overlap_geometry, net.batch_geometry = keep, keep_b
        # ---- dominant kernel: LFA backward at level 2, ch = 64
        ch, n2, D, ms_iso, ms_self = time_lfa_bwd(net.block2.lfa2, 1, geo)
        # primary figure: the launches inside training steps; the isolated launch on random operands stays beside it (it runs
        # 5-8 % slower: random activations / gradients draw more power than a net's — DVFS, MI355X_MICROARCH.md)
        ms = in_step["ms"] if (in_step and "ms" in in_step) else ms_iso
        # ALGORITHMIC flops of this backward (backward = 2 x forward: dF = dA W and dW = dA^T F, plus the
        # encoder's two transposes) vs the flops the kernel EXECUTES (it recomputes the forward attention GEMM
        # A = F W^T instead of saving [E, ch] logits: a third GEMM)
        flop_alg = 2 * 2 * n2 * K * (ch * ch + 10 * D)
        flop_exe = 3 * 2 * n2 * K * (ch * ch + 10 * D)
        tf = flop_alg / (ms * 1e-3) / 1e12
        tf_exe = flop_exe / (ms * 1e-3) / 1e12
        out["dominant"] = {"kernel": f"lfa_bwd_kernel<64,16,PIPE,fp32,FULL> (block2.lfa2, ch={ch}, n={n2}, K={K}), launched as in the "
                                     "step (partial-sum reduce deferred to the end of the backward pass)",
                           "bound": "mfma", "achieved": round(tf, 1), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                           "frac": round(tf / FP32_MFMA_PEAK_TF, 4),
                           "frac_algorithmic": round(tf / FP32_MFMA_PEAK_TF, 4),
                           "frac_executed": round(tf_exe / FP32_MFMA_PEAK_TF, 4),
                           **_traffic_fields("void lfa_bwd_kernel<64, 16, true"),
                           "algorithmic_flop_per_launch": flop_alg, "executed_flop_per_launch": flop_exe,
                           "avg_launch_ms": round(ms, 4),
                           "timed": ("HIP events around the layer's launch inside eagerly launched training steps (fwd + CE + bwd on "
                                     "the bench's batch), launch stream" if (in_step and "ms" in in_step) else
                                     "HIP events around an isolated launch on random operands"),
                           "in_step": in_step, "isolated_launch_ms": round(ms_iso, 4),
                           "frac_isolated_launch": round(flop_alg / (ms_iso * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4),
                           # rounds 1-4 timed the self-contained call (two memsets + this layer's partial-sum reduce behind
                           # the kernel): kept beside it for continuity
                           "self_contained_call_ms": round(ms_self, 4),
                           "frac_self_contained_call": round(flop_alg / (ms_self * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)}
        # round 6: the ch = 64 launch is the one with the largest share of the step, but not the slowest of
        # its family against the roofline — the three big launches side by side, `frac` of the entry = their MINIMUM
        fam = (in_step or {}).get("family") or {}
        if fam:
            rows = []
            for cc_, v in sorted(fam.items()):
                fa = 2 * 2 * v["n"] * K * (cc_ * cc_ + 10 * (cc_ // 2))
                rows.append({"kernel": f"lfa_bwd_kernel<{cc_},16,...,fp32,FULL>", "n": v["n"], "ch": cc_, "avg_launch_ms": round(v["ms"], 4),
                             "algorithmic_flop_per_launch": fa,
                             "frac": round(fa / (v["ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)})
            out["dominant"]["family_in_step"] = rows
            worst = min(rows, key=lambda r: r["frac"])
            if worst["ch"] != 64:
                # the headline fields describe the launch FURTHEST from the roofline; the ch = 64 launch (largest share of the
                # step: what rounds 1-5 reported) moves to "largest_share_launch"
                d = out["dominant"]
                d["largest_share_launch"] = {k: d[k] for k in ("kernel", "achieved", "frac", "frac_executed", "avg_launch_ms",
                                                               "algorithmic_flop_per_launch", "executed_flop_per_launch", "traffic")
                                             if k in d}
                tfw = worst["algorithmic_flop_per_launch"] / (worst["avg_launch_ms"] * 1e-3) / 1e12
                d.update({"kernel": f"{worst['kernel']} (block{2 + (worst['ch'] // 128)}.lfa2 ... ch={worst['ch']}, n={worst['n']}, K={K}): the launch of "
                                    "the family furthest from its roofline inside training steps",
                          "achieved": round(tfw, 1), "frac": worst["frac"], "frac_algorithmic": worst["frac"],
                          "frac_executed": round(1.5 * worst["frac"], 4), "avg_launch_ms": worst["avg_launch_ms"],
                          "algorithmic_flop_per_launch": worst["algorithmic_flop_per_launch"],
                          "executed_flop_per_launch": worst["algorithmic_flop_per_launch"] * 3 // 2})
                d.update(_traffic_fields(f"void lfa_bwd_kernel<{worst['ch']}, 16"))
        # ---- kNN + LSE gather stage: the K-NN tables of all four levels, both LFA layers of level 1 fwd + bwd
        stage = []
        for lvl in range(4):
            ix = geo.index[lvl]
            n = ix.n
            idx = torch.empty((n, K), dtype=torch.int32, device=dev)
            ms_knn = _time_launch(lambda: ops.call(
                "m3d_knn_query", ix.ws.data_ptr(), ix.ptr.data_ptr(), n, ix.num_clouds, None, 0, ix.ws.data_ptr(),
                ix.ptr.data_ptr(), n, K, 1, idx.data_ptr(), None, st))
            ent = hbm_entry(f"knn_query (self-kNN, level {lvl + 1}, n={n}, K={K})", n * (12 + 4 * K), ms_knn,
                            KNN_KERNEL_PREFIX[0 if lvl == 0 else 1], ((n + 63) // 64 * 64, (n + 255) // 256 * 256))
            # least VALU work of an exact K-NN query on a grid: ~2 K candidates inspected (a disc holding K points sits in a
            # square of ~2 K), 8 lane operations per distance + 2 per comparison, K log2 K compare-exchanges to keep the list
            ent.update(_valu_fields(KNN_KERNEL_PREFIX[0 if lvl == 0 else 1], n, ms_knn, 2 * K * 10 + K * 4 * 2, "query",
                                    ((n + 63) // 64 * 64, (n + 255) // 256 * 256)))
            ent["binding_roofline"] = "valu"
            stage.append(ent)
        for lfa in (net.block1.lfa1, net.block1.lfa2):
            ch1, n1, ms_f = time_lfa_fwd(lfa, 0, geo)
            ent = hbm_entry(f"lfa_fwd_full_kernel<{ch1},16> (level 1, ch={ch1}, n={n1}, K={K})",
                            n1 * (12 + 4 * ch1 // 2 + 4 * K + 4 * ch1), ms_f, f"void lfa_fwd_full_kernel<{ch1}, 16")
            # least VALU work per edge: relative position (9), encoder ch/2 x (10 FMA + 1 activation), softmax-weighted sum
            # ch x (scale, exp, 2 accumulations); the attention product itself runs on the matrix instruction
            ent.update(_valu_fields(f"void lfa_fwd_full_kernel<{ch1}, 16", n1 * K, ms_f, 9 + (ch1 // 2) * 11 + ch1 * 4, "edge"))
            ent["binding_roofline"] = "valu"
            stage.append(ent)
        for lfa in (net.block1.lfa1, net.block1.lfa2):
            edge_rows = bool(ops.USE_LFA_EDGE_ROWS and ops.USE_LFA_FULL and
                             ops.lib().m3d_lfa_bwd_edge_rows_ok(plan.totals[0], K, lfa.mlp_attention.lins[0].weight.shape[0],
                                                                ops.LRELU_SLOPE))
            if edge_rows:
                ch1, n1, D1, ms_b, ms_k = time_lfa_bwd_edge_rows(lfa, 0, geo)
                name = f"lfa_bwd_small_kernel<{ch1},edge rows> + gather_sum_rows4 (level 1, ch={ch1}, n={n1}, K={K})"
                prefix = f"void lfa_bwd_small_kernel<{ch1}, true"
            else:
                ch1, n1, D1, ms_b, _ = time_lfa_bwd(lfa, 0, geo)
                ms_k = ms_b
                name, prefix = f"lfa_bwd_kernel<{ch1},16> (level 1, ch={ch1}, n={n1}, K={K})", f"void lfa_bwd_kernel<{ch1}, 16"
            ent = hbm_entry(name, n1 * (12 + 4 * D1 + 4 * ch1 + 4 * K) + 4 * n1 * D1, ms_b, prefix)
            # forward recompute (as above) + softmax backward ch x 6 + activation derivative / scatter ch x 2 per edge
            ent.update(_valu_fields(prefix, n1 * K, ms_k, 9 + (ch1 // 2) * 11 + ch1 * 4 + ch1 * 8, "edge"))
            if edge_rows:
                ent["kernel_alone_ms"] = round(ms_k, 4)
                ent["binding_roofline"] = ("valu (kernel) + hbm (the [n K, D] edge rows written once and read once by the "
                                           "reverse-list gather: counted in `traffic` of the kernel, not in the algorithmic bytes)")
            else:
                ent["binding_roofline"] = "valu + dx atomics"
            stage.append(ent)
        out["knn_lse"] = stage
    return out


# kernel-name prefixes of the self-kNN query in the PMC passes (level 1 takes the deferred-insertion kernel, the
# deeper levels the direct one); the CSV has one row per launch grid (= queries rounded up to 64)
KNN_KERNEL_PREFIX = ("void knn_query_queue_kernel<16", "void knn_query_kernel<16")


def _pick_threads():
    """Thread count for the CPU baseline: the fastest of {1, 4, 8, 16, 32, 64, 128, all cores} on a micro-probe shaped like
    the oracle's level-1 edge tensors (some hosts — e.g. oversubscribed VMs — are slower with every core)."""
    ncpu = os.cpu_count() or 1
    a, b, w = torch.rand(204800, 16), torch.rand(204800, 16), torch.rand(16, 16)
    best, best_t = 1, float("inf")
    for th in sorted({1, min(4, ncpu), min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu), min(128, ncpu), ncpu}):
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
    path), BASELINE.md section 3 protocol: fwd+bwd in train mode (CE loss) and fwd-only in eval mode, 3 warm-up + 10
    timed iterations, median, on ALL of the GPU line's 16 tiles (the same batch the GPU steps on), at the thread count a
    micro-probe picks (the one deviation from that section, stated in the line: on the 256-core GPU host "all cores" did
    not finish three iterations in 460 s).  ~2 minutes of CPU time at 16 threads.  ``full``: the same once more on every
    host core."""
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

    def leg(threads, warm, reps, only_train=False):
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
            if only_train and name != "fwd_bwd":
                continue
            for _ in range(warm):
                fn()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            res[name] = tiles * points / statistics.median(ts)
        return res

    warm, reps = 3, 10  # BASELINE.md section 3 (rounds 3-4 ran 1 + 3 by default: VERDICT r4, "protocol drift")
    _progress(f"CPU baseline: {tiles} tiles, {picked} threads, {warm} + {reps} iterations")
    main = leg(picked, warm, reps)
    out = {"value": round(main["fwd_bwd"], 1), "unit": "points/s", "cores": picked, "kind": "port",
           "fwd_only": round(main["fwd_only"], 1), "host_cores": ncpu,
           "sample": f"{tiles} tiles x {points} pts (the GPU line's whole batch), median of {reps} timed iterations after {warm} "
                     f"warm-up (BASELINE.md section 3's protocol); fwd+bwd = train mode + CE loss + backward, fwd_only = eval / no_grad; "
                     f"oracle/randla_oracle.py (unfused torch CPU ops, cKDTree kNN); threads = {picked} (fastest of "
                     f"{{1,4,8,16,32,64,128,{ncpu}}} on a micro-probe), host has {ncpu} cores"}
    if full and picked != ncpu:
        # every host core: on the 256-core GPU host the oversubscribed run did not finish 3 iterations of 2 tiles in 460 s
        # (profiles/r03f_bench.err) — only on request
        _progress(f"CPU baseline: all {ncpu} cores")
        allc = leg(ncpu, 3, 10, False)
        out["all_cores"] = {"cores": ncpu, "value": round(allc["fwd_bwd"], 1), "fwd_only": round(allc["fwd_only"], 1),
                            "sample": "same tiles, 3 + 10"}
    return out


def torch_rocm_baseline(dev, tiles, points, K):
    """BASELINE.md section 3's second, informative baseline: the restated reference path (the oracle's unfused module
    tree, op for op) through STOCK PyTorch-ROCm ops on the same MI355X — gathers, cat, Linear, BatchNorm1d,
    scatter-add, segment softmax, kNN by torch.cdist + topk per cloud, Python-loop decimation with host syncs like
    pyg_randla_net.py:219-229 — eval forward and train step (CE + backward, no optimizer).  What the hand-written
    kernels buy over "just run the PyG model on the GPU" (torch_cluster's CUDA kNN is a brute force as well)."""
    import statistics

    from oracle.randla_oracle import RandLANetOracle
    from myria3d_amd.synthetic import synthetic_batch

    torch.manual_seed(0)
    net = RandLANetOracle(9, 6, num_neighbors=K, return_logits=True, knn="cdist").to(dev)
    x, pos, batch, ptr, y = synthetic_batch([points] * tiles)
    x, pos, y = x.to(dev), pos.to(dev), y.to(dev)  # (ptr stays on the host: the oracle reads it with int())

    def train_step():
        net.train()
        net.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(net(x, pos, None, ptr), y).backward()

    def fwd_step():
        net.eval()
        with torch.no_grad():
            net(x, pos, None, ptr)

    res = {}
    for name, fn in (("fwd_bwd", train_step), ("fwd_only", fwd_step)):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[name] = statistics.median(ts)
    n = tiles * points
    return {"value": round(n / res["fwd_bwd"], 1), "unit": "points/s", "ms_per_step": round(res["fwd_bwd"] * 1e3, 2),
            "fwd_only": round(n / res["fwd_only"], 1), "fwd_only_ms": round(res["fwd_only"] * 1e3, 2),
            "what": f"oracle module tree on cuda through stock torch-ROCm ops (cdist + topk kNN), {tiles} x {points} pts, "
                    "fp32, eager, median of 5 after 3 warm-up; fwd+bwd = train mode + CE + backward (no optimizer)"}


def dropin_bench(args, dev):
    """``--mode dropin``: the step a Lightning loop gets by changing ONE yaml key (model.py:61-62,79,105-120): the
    class behind the boundary, nothing else — ``HipRandLANet(x, pos, batch, ptr)`` with no plan, no
    ``prefetch_geometry``, no hipGraph, parameters NOT flattened, ``torch.nn.CrossEntropyLoss(ignore_index=65)`` and
    ``torch.optim.Adam`` (a process of its own: hardware queues are shared inside one)."""
    from myria3d_amd import HipRandLANet
    from myria3d_amd.synthetic import synthetic_batch

    B, N, K = args.tiles, args.points, args.neighbors
    x, pos, batch, ptr, y = synthetic_batch([N] * B)
    x, pos, batch, ptr, y = (t.to(dev) for t in (x, pos, batch, ptr, y))
    torch.manual_seed(0)
    net = HipRandLANet(9, 6, decimation=4, num_neighbors=K, return_logits=True).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=0.003933709606504788)
    crit = torch.nn.CrossEntropyLoss(ignore_index=65)

    def step():
        opt.zero_grad()
        crit(net(x, pos, batch, ptr), y).backward()
        opt.step()

    for _ in range(30):
        step()
    dt = timed(step, 15, 1) / 15
    net.eval()

    def fwd():
        with torch.no_grad():
            net(x, pos, batch, ptr)

    for _ in range(10):
        fwd()
    dtf = timed(fwd, 15, 1) / 15
    # the same step when EVERY batch has its own tile sizes, as Lightning's loader delivers them (points_budget.yaml:
    # 300 ... 40 000 nodes per tile): 8 batches of B tiles with sizes drawn around N, cycled — plans, arena and
    # geometry buffers see new shapes every step
    import numpy as np

    net.train()
    rs = np.random.RandomState(0)
    var, var_sizes = [], []
    for i in range(8):
        sizes = [int(v) for v in rs.randint(N // 2, N + N // 2 + 1, size=B)]
        vx, vpos, vbatch, vptr, vy = synthetic_batch(sizes, first_tile_id=100 * i)
        var.append(tuple(t.to(dev) for t in (vx, vpos, vbatch, vptr, vy)))
        var_sizes.append(sizes)
    turn = [0]

    def vstep():
        vx, vpos, vbatch, vptr, vy = var[turn[0] % len(var)]
        turn[0] += 1
        opt.zero_grad()
        crit(net(vx, vpos, vbatch, vptr), vy).backward()
        opt.step()

    for _ in range(16):
        vstep()
    dtv = timed(vstep, 16, 1) / 16
    mean_pts = sum(int(v[3][-1]) for v in var) / len(var)
    # ... and with the opt-ins of INTEGRATION.md section 3 on the same changing layouts: flat buffers + FusedAdam + the HIP
    # criterion + the next batch's position-only work interleaved with this step (no graph: the layouts differ)
    from myria3d_amd import FusedAdam, cross_entropy

    del opt
    net2 = HipRandLANet(9, 6, decimation=4, num_neighbors=K, return_logits=True).to(dev).train().flatten_parameters()
    opt2 = FusedAdam(net2, lr=0.003933709606504788)
    turn[0] = 0

    def ostep():
        vx, vpos, vbatch, vptr, vy = var[turn[0] % len(var)]
        nxt = var[(turn[0] + 1) % len(var)]
        turn[0] += 1
        net2.prefetch_geometry(nxt[1], nxt[3], interleave=True)
        cross_entropy(net2(vx, vpos, vbatch, vptr), vy, ignore_index=65).backward()
        opt2.step()

    net2.prefetch_geometry(var[0][1], var[0][3])
    for _ in range(24):
        ostep()
    dto = timed(ostep, 16, 1) / 16
    net2.join_geometry()
    # ... and when the loop hands over the tile sizes it already holds on the HOST (a loader collates on the CPU:
    # ``batch.ptr.tolist()`` before the transfer costs nothing): the plan of every batch is built from
    # them (a fresh ``make_plan`` per step, uploaded asynchronously) and nothing in the step reads the device — the host
    # runs ahead of the GPU instead of waiting for the previous step at every ``ptr.tolist()``
    from myria3d_amd import make_plan

    host_ptr = [[0] + list(np.cumsum(sz_)) for sz_ in var_sizes]
    live = {}

    def hstep():
        i, j = turn[0] % len(var), (turn[0] + 1) % len(var)
        vx, vpos, vbatch, vptr, vy = var[i]
        turn[0] += 1
        live[j] = make_plan(host_ptr[j], 4, K, dev)
        net2.prefetch_geometry(var[j][1], var[j][3], live[j], interleave=True)
        cross_entropy(net2(vx, vpos, vbatch, vptr, plan=live.pop(i)), vy, ignore_index=65).backward()
        opt2.step()

    def hprime():
        net2.join_geometry()
        live.clear()
        live[turn[0] % len(var)] = make_plan(host_ptr[turn[0] % len(var)], 4, K, dev)
        net2.prefetch_geometry(var[turn[0] % len(var)][1], var[turn[0] % len(var)][3], live[turn[0] % len(var)])

    hprime()
    for _ in range(16):
        hstep()
    dth = timed(hstep, 16, 1) / 16
    net2.join_geometry()
    net2.prefetch_geometry(var[turn[0] % len(var)][1], var[turn[0] % len(var)][3])
    # ... and with the backward pass on the calling thread (one line at program start; one device per
    # process leaves the autograd engine's device thread nothing to overlap, and the hand-off costs ~10 us per node)
    torch.autograd.set_multithreading_enabled(False)
    for _ in range(8):
        ostep()
    dto_st = timed(ostep, 16, 1) / 16
    hprime()
    for _ in range(8):
        hstep()
    dth_st = timed(hstep, 16, 1) / 16
    net2.join_geometry()
    opt3 = torch.optim.Adam(net.parameters(), lr=0.003933709606504788)

    def vstep_st():
        vx, vpos, vbatch, vptr, vy = var[turn[0] % len(var)]
        turn[0] += 1
        opt3.zero_grad()
        crit(net(vx, vpos, vbatch, vptr), vy).backward()
        opt3.step()

    for _ in range(8):
        vstep_st()
    dtv_st = timed(vstep_st, 16, 1) / 16
    torch.autograd.set_multithreading_enabled(True)
    print(json.dumps({"dropin_eager_ms_per_step": round(dt * 1e3, 4), "value": round(B * N / dt, 1),
                      "optin_variable_layout_ms_per_step": round(dto * 1e3, 4),
                      "optin_variable_layout_points_per_s": round(mean_pts / dto, 1),
                      "dropin_variable_layout_ms_per_step": round(dtv * 1e3, 4),
                      "dropin_variable_layout_points_per_s": round(mean_pts / dtv, 1),
                      "optin_variable_layout_single_thread_autograd_ms_per_step": round(dto_st * 1e3, 4),
                      "dropin_variable_layout_single_thread_autograd_ms_per_step": round(dtv_st * 1e3, 4),
                      "optin_host_sizes_variable_layout_ms_per_step": round(dth * 1e3, 4),
                      "optin_host_sizes_variable_layout_single_thread_autograd_ms_per_step": round(dth_st * 1e3, 4),
                      "variable_layout": f"8 batches of {B} tiles, sizes uniform in [{N // 2}, {N + N // 2}] (mean "
                                         f"{mean_pts:.0f} points per batch), a different layout every step",
                      "fwd_only_ms": round(dtf * 1e3, 4), "unit": "points/s",
                      "what": "HipRandLANet.forward(x, pos, batch, ptr) + torch CrossEntropyLoss + backward + torch.optim.Adam, "
                              "eager, no plan / prefetch_geometry / hipGraph / flat buffers"}), flush=True)


def pointnet2_bench(args, dev):
    """``--mode pointnet2``: BASELINE configs[4]'s second half — the PointNet++ set-abstraction variant
    (``myria3d_amd.pointnet2.HipPointNet2``: farthest-point sampling, kNN grouping, SharedMLP over the edge rows, max
    aggregation, FPModule decoder; no reference implementation exists, model.py:12) — one eager training step
    (forward + CrossEntropy + backward + torch Adam) and the eval forward, with the sampler's share timed beside."""
    from myria3d_amd import ops
    from myria3d_amd.pointnet2 import HipPointNet2
    from myria3d_amd.synthetic import synthetic_batch

    B, N, K = args.tiles, args.points, args.neighbors
    x, pos, batch, ptr, y = synthetic_batch([N] * B)
    x, pos, batch, ptr, y = (t.to(dev) for t in (x, pos, batch, ptr, y))
    torch.manual_seed(0)
    net = HipPointNet2(9, 6, decimation=4, num_neighbors=K, return_logits=True).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=0.003933709606504788)
    crit = torch.nn.CrossEntropyLoss(ignore_index=65)

    def step():
        opt.zero_grad()
        crit(net(x, pos, batch, ptr), y).backward()
        opt.step()

    def step_pipelined():
        # the next step's position-only work (farthest-point sampling: a serial chain on 16 of the 256 CUs, kNN grids, grouping
        # and 1-NN tables) is enqueued on the side streams in front of this step's forward and runs under the whole step
        # (HipPointNet2.prefetch_geometry, round 5)
        opt.zero_grad()
        net.prefetch_geometry(pos, ptr, wait_main=False)  # the NEXT step's tables (the batch is resident): queued behind this step's
        crit(net(x, pos, batch, ptr), y).backward()      # (this forward takes the tables queued one step ago)
        opt.step()

    steps = max(2, min(args.steps, 10))
    for _ in range(3):
        step()
    dt_serial = timed(step, steps, 1) / steps
    # round 6: several batches' position-only work in flight, each on its own stream pair (with three, a sampler chain
    # completes every ~11 ms: the step is bound by the feature kernels).  The class default (3) is measured FIRST: every
    # depth adds stream pairs to the process, and which hardware queue a later stream lands on decides what overlaps
    by_depth = {}
    for depth in (3, 2, 4):  # (2 = ONE batch ahead, round 5: the step is then bound by the sampler's latency)
        net.prefetch_depth = depth
        for _ in range(depth - 1):
            net.prefetch_geometry(pos, ptr)
        for _ in range(depth + 1):
            step_pipelined()
        by_depth[depth] = timed(step_pipelined, steps, 1) / steps
        net._look = None
        torch.cuda.synchronize()
    dt_depth1 = by_depth[2]
    dt = by_depth[3]  # (the class default)
    net._look = None
    net.eval()

    def fwd():
        with torch.no_grad():
            net(x, pos, batch, ptr)

    fwd()
    dtf = timed(fwd, steps, 1) / steps
    plan = net.plan_for(ptr)
    pos4 = ops.pad_pos(pos)
    lv = [pos4]
    for l in range(3):
        lv.append(ops.gather_rows(lv[l], net.last_sample_idx[l]))

    ixs = [ops.KnnIndex(lv[l], plan.ptrs[l]) for l in range(3)]

    def sampler():  # what the net runs: exact bucket skipping above 16 384 points per cloud, the register-resident sampler below
        for l in range(3):
            ops.fps(lv[l], plan.ptrs[l], plan.ptrs[l + 1], plan.totals[l + 1], plan.max_points[l], index=ixs[l])

    dts = timed(sampler, steps, 1) / steps

    def sampler_single():  # round 3's sampler: every point in every iteration
        for l in range(3):
            ops.fps(lv[l], plan.ptrs[l], plan.ptrs[l + 1], plan.totals[l + 1], plan.max_points[l])

    dts1 = timed(sampler_single, max(1, steps // 2), 1) / max(1, steps // 2)
    print(json.dumps({"metric": "points/sec fwd+bwd, PointNet++ set-abstraction variant", "value": round(B * N / dt, 1),
                      "unit": "points/s", "ms_per_step": round(dt * 1e3, 3), "serial_ms_per_step": round(dt_serial * 1e3, 3),
                      "one_batch_ahead_ms_per_step": round(dt_depth1 * 1e3, 3), "prefetch_depth": 3,
                      "ms_per_step_by_prefetch_depth": {str(k): round(v * 1e3, 3) for k, v in sorted(by_depth.items())},
                      "fwd_only_ms": round(dtf * 1e3, 3),
                      "fps_ms": round(dts * 1e3, 3), "fps_plain_ms": round(dts1 * 1e3, 3), "dtype": "f32", "data": "synthetic",
                      "workload": f"HipPointNet2 train step, {B} tiles x {N} pts, K={K}, decimation 4, FPS sampling, eager "
                                  "launches, torch Adam (BASELINE configs[4], second half; no reference implementation: "
                                  "oracle-only parity)",
                      "pipelining": "ms_per_step: the next step's position-only work (sampler, grids, grouping / 1-NN tables) runs one step "
                                    "ahead on a side stream (HipPointNet2.prefetch_geometry); serial_ms_per_step: everything inside the step",
                      "what": "fps_ms = the three farthest-point-sampling launches of one forward (one workgroup per tile; exact "
                              "bucket skipping over the kNN grid's cell-sorted records above 16 384 points per tile); "
                              "fps_plain_ms: every point visited in every iteration (round 3's sampler)"}), flush=True)


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


def predict_e2e_inputs(dev, side_m=1000.0, density=10.0):
    """(net, pos, x): the synthetic cloud of ``predict_e2e_bench`` resident on the device and a random-init eval-mode net."""
    import numpy as np

    from myria3d_amd import HipRandLANet

    rs = np.random.RandomState(0)
    n = int(side_m * side_m * density)
    xy = rs.uniform(0, side_m, (n, 2)).astype(np.float32)
    z0 = 2.0 * np.sin(2 * np.pi * xy[:, 0] / 50.0) + 1.5 * np.cos(2 * np.pi * xy[:, 1] / 37.0)
    u = rs.uniform(size=n).astype(np.float32)
    z = np.where(u < 0.5, z0 + 0.05 * rs.standard_normal(n), np.where(u < 0.85, z0 + 15 * rs.uniform(size=n), z0 + 3 + 6 * rs.uniform(size=n)))
    pos = torch.from_numpy(np.concatenate([xy, z[:, None].astype(np.float32)], 1)).to(dev)
    x = torch.rand((n, 9), device=dev)
    x[:, 0] = torch.from_numpy(rs.gamma(2.0, 300.0, n).astype(np.float32)).to(dev)
    x[:, 7] = x[:, 7] * 255.0
    torch.manual_seed(0)
    net = HipRandLANet(9, 7, num_neighbors=16, return_logits=True).to(dev).eval()
    return net, pos, x


def predict_e2e_bench(args, dev, side_m=1000.0, density=10.0, reps=2):
    """BASELINE config 3 END TO END: one synthetic 1 km^2 cloud (10 M points at 10 pts/m^2; raw Lambert-style coordinates minus a
    file offset, raw Intensity / colour features) in HBM through the whole ``predict.py`` chain on the device —
    ``myria3d_amd.predict_cloud``: tile selection (400 samples of 50 m) -> GridSampling(0.25) -> node budget -> Center /
    NullifyLowestZ / NormalizePos / StandardizeRGBAndIntensity -> forward (batches of 50 samples) -> knn_interpolate(k=10) onto
    every original point -> scatter_sum merge -> softmax / argmax / entropy.  The timed call starts from the resident cloud
    and ends with the per-point predictions on the device (LAS reading / writing is pdal's: storage, out of scope)."""
    from myria3d_amd import predict_cloud

    net, pos, x = predict_e2e_inputs(dev, side_m, density)
    n = pos.shape[0]
    out = predict_cloud(net, pos, x, tile_width=side_m, subtile_width=50, batch_size=50)  # warm-up (allocator, plans)
    # every point is predicted (a point exactly ON a sample border belongs to both samples — the reference's closed ball — and
    # is predicted twice, its logits summed: a handful among 10 M fp32 coordinates)
    dup = out["idx_in_full_cloud"].numel() - n
    assert 0 <= dup < 1000 and int(out["idx_in_full_cloud"].unique().numel()) == n
    assert bool(torch.isfinite(out["probas"]).all()) and float((out["probas"].sum(1) - 1).abs().max()) < 1e-4
    del out
    dt = timed(lambda: predict_cloud(net, pos, x, tile_width=side_m, subtile_width=50, batch_size=50), reps, 1) / reps
    return {"value": round(n / dt, 1), "unit": "points/s", "ms_per_cloud": round(dt * 1e3, 1), "points": n,
            "points_on_sample_borders_predicted_twice": dup,
            "workload": f"BASELINE config 3 end to end: {n} points over {side_m:.0f} m x {side_m:.0f} m, 400 samples of 50 m, batch 50, "
                        "GridSampling 0.25 m, K=16, C=7, interpolation k=10 (myria3d_amd.predict_cloud)"}


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
    forwards over B tiles x N points per rank, launched through ``myria3d_amd.GraphedStep`` (the launch form is product
    code with its own parity tests).  Returns (record, net, pos, plan)."""
    from myria3d_amd import FusedAdam, GraphedStep, HipRandLANet
    from myria3d_amd.ddp import broadcast_module_state, shard_tiles
    from myria3d_amd.synthetic import synthetic_batch

    tile_ids = shard_tiles(B * world, rank, world)  # weak scaling: B tiles per rank
    x, pos, batch, ptr, y = synthetic_batch([N] * B, first_tile_id=tile_ids.start)
    x, pos, ptr, y = x.to(dev), pos.to(dev), ptr.to(dev), y.to(dev)
    torch.manual_seed(0)
    net = HipRandLANet(9, 6, decimation=4, num_neighbors=K, return_logits=True).to(dev)
    net.matmul_precision = "bf16" if precision in ("bf16", "bf16ops") else precision
    if precision == "bf16":
        net.activation_dtype = torch.bfloat16  # every feature matrix and its gradient in HBM as bf16 (round 6)
    # every parameter / gradient becomes a view of one flat buffer: the backward kernels accumulate into it, RCCL
    # all-reduces it as ONE 4.45 MB bucket, m3d_adam_step updates (and clears) it in one launch
    net.flatten_parameters()
    broadcast_module_state(net)
    opt = FusedAdam(net, lr=0.003933709606504788, all_reduce=True,  # lr: configs/model/pyg_randla_net_model.yaml:4
                    force_collective=args.force_collective)
    look = args.lookahead
    mode = "eager" if args.no_graph else args.launch

    def make(kind, launch):
        gs = GraphedStep(net, ptr, x.shape[1], mode=kind, optimizer=opt if kind == "train" else None, ignore_index=65,
                         lookahead=look, launch=launch, lookahead_mode=args.lookahead_mode,
                         collective=args.collective)
        gs.load_all(x, pos, y)
        made[kind, launch] = gs
        return gs

    made = {}

    def pick(kind, eager_warm, graph_warm):
        """(step function, launch name, probe timings): the replayed hipGraphs, the eager launcher, or (``auto``)
        whichever a short probe of each measures faster on this host — both run the same kernels."""
        probe, fn, launch = {}, None, "eager"
        if with_eager or mode in ("auto", "eager"):
            ge = make(kind, "eager")
            for _ in range(eager_warm):  # (the eager path needs ~30 steps to settle: allocator, arena, geometry slots)
                ge.step()
            probe["eager"] = timed(ge.step, 15, world) / 15 * 1e3
            fn = ge.step
        if mode != "eager":
            try:
                gg = make(kind, "graph").prepare(preserve_state=False)
                for _ in range(graph_warm):
                    gg.step()
                probe["hipgraph"] = timed(gg.step, 10, world) / 10 * 1e3
                if mode == "graph" or fn is None or probe["hipgraph"] <= probe["eager"]:
                    fn, launch = gg.step, "hipgraph"
            except Exception as e:  # capture is an optimisation, never a requirement
                if rank == 0:
                    print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                torch.cuda.synchronize()
                if fn is None:
                    fn = make(kind, "eager").step
        return fn, launch, probe

    _progress("train step: launch probe")
    step_fn, launch, probe = pick("train", 40, 5)
    _progress(f"train step: timed region ({launch})")
    for _ in range(warmup):
        step_fn()
    dt = timed(step_fn, steps, world)
    rank_seconds = LAST_RANK_SECONDS if world > 1 else None
    # eval forward of the trained weights (the first eval pass folds the BatchNorms / packs the attention weights; the
    # module caches them until the next training phase)
    _progress("eval forward")
    fwd_fn, flaunch, fprobe = pick("eval", 20, 3)
    for _ in range(max(1, warmup // 2)):
        fwd_fn()
    dt_f = timed(fwd_fn, steps, world)

    total_points = B * N * world
    collective = opt.uses_collective()
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
        "dtype": {"fp32": "f32", "bf16": "bf16 activation storage + bf16 matrix-core operands; f32 accumulate, parameters, "
                                         "statistics (f64 sums), positions, logits",
                  "bf16ops": "f32 storage / accumulate, bf16 matrix-core operands",
                  "bf16x3": "f32 storage / accumulate; attention GEMMs of the LFA layers with >= 64 channels as split-bf16 products "
                            "(hi + lo operands, three bf16 matrix-core products); everything else f32"}[precision],
        "data": "synthetic",
        "config": {"workload": f"RandLA-Net train step (fwd + CE + bwd + grad all-reduce + Adam), {B} tiles x {N} pts per GPU, "
                               f"K={K}, F=9, C=6, decimation 4 ({_baseline_config(N, K)}, {precision})",
                   "tiles_per_gpu": B, "points_per_tile": N, "num_neighbors": K, "parallelism": f"dp{world} over tiles",
                   "collective": ("one flat 4.45 MB fp32 gradient all-reduce per step (RCCL)" +
                                  (" — forced on a 1-rank group" if world == 1 else "") +
                                  (f"; {made['train', 'graph'].collective} in the step's hipGraph form" if ("train", "graph") in made and launch == "hipgraph" else "")
                                  ) if collective else "none (1 rank)",
                   "launch": launch + (" (myria3d_amd.GraphedStep)"), **({"geometry_lookahead": args.lookahead_mode} if look else {}),
                   **({"side_stream_candidates_ms": made["train", "graph"].side_stream_ms} if ("train", "graph") in made else {})},
        "fwd_only": {"value": round(total_points * steps / dt_f, 1), "unit": "points/s",
                     "ms_per_step": round(dt_f / steps * 1e3, 4), "mode": "eval, no_grad", "launch": flaunch},
    }
    if rank_seconds:
        # every rank's own clock over the same timed steps (before the closing barrier): a slow rank shows here
        per = [round(v / steps * 1e3, 4) for v in rank_seconds]
        res["per_rank_ms_per_step"] = {"min": min(per), "max": max(per), "ranks": per}
    if probe:
        res["launch_probe_ms"] = {k: round(v, 4) for k, v in probe.items()}
    if "eager" in probe:
        res["eager_ms_per_step"] = round(probe["eager"], 4)  # same kernels launched one by one, lookahead interleaved
    if collective:
        # the gradient exchange alone: ONE all-reduce of the flat bucket, HIP events on the current stream
        g = net.flat_grads
        for _ in range(3):
            dist.all_reduce(g)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a_, b_ in evs:
            a_.record()
            dist.all_reduce(g)
            b_.record()
        torch.cuda.synchronize()
        g.zero_()
        res["allreduce_ms"] = round(sum(a_.elapsed_time(b_) for a_, b_ in evs) / len(evs), 4)
        res["allreduce_bytes"] = g.numel() * 4
    from myria3d_amd import make_plan

    return res, net, pos, make_plan(ptr.tolist(), 4, K, dev)


def _extra_legs(args, dev, res, B, N, K):
    """Informative legs of the N = 1 line (BASELINE configs 3 and 5, the bf16 mode, the plain drop-in step, the N > 1 code
    path on a 1-rank RCCL group, the stock-torch baseline): short, they share the driver's clock; ``--skip-legs`` drops any."""
    skip = set(filter(None, args.skip_legs.split(",")))

    def leg(name, key, fn):
        if name in skip:
            return
        try:
            torch.cuda.empty_cache()
            fn()
        except Exception as e:
            res[key] = {"error": f"{type(e).__name__}: {e}"}

    def predict():
        # (a process of its own, like the other legs: behind the training leg in THIS process the chain's three streams share
        # hardware queues with the first net's and measure 52.5 instead of 48.8 ms per cloud)
        pr = _leg_in_fresh_process(["--mode", "predict"])
        res["predict_config3"] = {k: pr[k] for k in ("value", "unit", "ms_per_sweep")} | {"workload": pr["config"]["workload"]}
        res["predict_config3_end_to_end"] = pr["end_to_end"]

    def bf16():
        # BASELINE config 2 names bf16: the same step with the matrix-bound layers on bf16 matrix cores (fp32 accumulate;
        # parity bar of SURVEY 8c: logits within 3e-2 of the fp32 oracle, tests/test_gpu_net.py)
        b16 = _leg_in_fresh_process(["--precision", "bf16", "--steps", str(args.steps), "--warmup", str(args.warmup),
                                     "--tiles", str(B), "--points", str(N), "--neighbors", str(K)])
        res["bf16"] = {"value": b16["value"], "unit": "points/s", "ms_per_step": b16["ms_per_step"],
                       "fwd_only": b16["fwd_only"],
                       "what": "round 6: every feature matrix and its gradient stored as bf16 (M3D_IO_BF16: GEMM / BatchNorm / "
                               "weight-gradient / LFA / row kernels read and write 2-byte elements, fp32 registers) + LFA "
                               "attention GEMMs (ch >= 64, fwd + bwd) and SharedMLP GEMMs with K > 64 on "
                               "v_mfma_f32_16x16x32_bf16; fp32: accumulate, parameters, statistics (fp64 sums), positions, kNN, "
                               "softmax, logits, parameter gradients, Adam"}
        _progress("bf16 operands only (fp32 storage: the leg of rounds 2-5)")
        bo = _leg_in_fresh_process(["--precision", "bf16ops", "--steps", str(args.steps), "--warmup", str(args.warmup),
                                    "--tiles", str(B), "--points", str(N), "--neighbors", str(K)])
        res["bf16_operands_only"] = {"value": bo["value"], "unit": "points/s", "ms_per_step": bo["ms_per_step"],
                                     "fwd_only": bo["fwd_only"], "what": "fp32 storage, bf16 matrix-core operands only"}

    def bf16x3():
        # round-5 experiment: the fp32 contract's attention GEMMs as split-bf16 products on the bf16 matrix
        # cores (the f32-input MFMA runs at the vector rate on the pipe the kernels' VALU phases need); meets the UNCHANGED fp32
        # tolerances (tests/test_gpu_net.py::test_split_bf16_mode_meets_the_fp32_tolerances) — its own dtype, not the headline
        b3 = _leg_in_fresh_process(["--precision", "bf16x3", "--steps", str(args.steps), "--warmup", str(args.warmup),
                                    "--tiles", str(B), "--points", str(N), "--neighbors", str(K)])
        res["bf16x3_split_products"] = {"value": b3["value"], "unit": "points/s", "ms_per_step": b3["ms_per_step"],
                                        "fwd_only": b3["fwd_only"], "dtype": b3["dtype"],
                                        "parity": "the fp32 contract's tolerances (eval logits 1e-4, every parameter gradient 5e-3)"}

    def dropin():  # the plain drop-in step (what model.py:79 + Lightning's loop get) in a process of its own
        di = _leg_in_fresh_process(["--mode", "dropin", "--tiles", str(B), "--points", str(N), "--neighbors", str(K)])
        res["dropin_eager_ms_per_step"] = di["dropin_eager_ms_per_step"]
        res["dropin_variable_layout_ms_per_step"] = di.get("dropin_variable_layout_ms_per_step")
        res["optin_variable_layout_ms_per_step"] = di.get("optin_variable_layout_ms_per_step")
        res["optin_variable_layout_single_thread_autograd_ms_per_step"] = di.get("optin_variable_layout_single_thread_autograd_ms_per_step")
        res["dropin_variable_layout_single_thread_autograd_ms_per_step"] = di.get("dropin_variable_layout_single_thread_autograd_ms_per_step")
        res["optin_host_sizes_variable_layout_ms_per_step"] = di.get("optin_host_sizes_variable_layout_ms_per_step")
        res["optin_host_sizes_variable_layout_single_thread_autograd_ms_per_step"] = di.get("optin_host_sizes_variable_layout_single_thread_autograd_ms_per_step")
        res["dropin"] = di

    def collective():  # RCCL on this box: the N > 1 code path on a 1-rank group (collective + capture interplay)
        fc = _leg_in_fresh_process(["--force-collective", "--steps", str(args.steps), "--warmup", str(args.warmup),
                                    "--tiles", str(B), "--points", str(N), "--neighbors", str(K)])
        res["forced_collective_1rank"] = {k: fc[k] for k in ("ms_per_step", "allreduce_ms", "allreduce_bytes", "rccl_ranks",
                                                             "launch_probe_ms") if k in fc} | {
            "launch": fc["config"]["launch"], "collective": fc["config"]["collective"]}
        # round 3's form of the same step (all-reduce + Adam after the graph), for the A/B
        fe = _leg_in_fresh_process(["--force-collective", "--collective", "eager", "--launch", "graph", "--steps", str(args.steps),
                                    "--warmup", str(args.warmup), "--tiles", str(B), "--points", str(N), "--neighbors", str(K)])
        res["forced_collective_1rank"]["eager_collective_ms_per_step"] = fe["ms_per_step"]

    def torch_leg():
        _progress("torch-ROCm baseline")
        res["torch_rocm_baseline"] = torch_rocm_baseline(dev, B, N, K)

    def dense():
        d5 = _leg_in_fresh_process(["--steps", "5", "--warmup", "2", "--tiles", "16", "--points", "40000", "--neighbors", "32"])
        res["dense_tiles_config5"] = {"value": d5["value"], "unit": "points/s", "ms_per_step": d5["ms_per_step"],
                                      "fwd_only": d5["fwd_only"], "workload": d5["config"]["workload"]}

    def pointnet2():
        # (10 timed steps: the closing synchronize also waits for the sampler chains queued for the NEXT steps — with 3 steps
        # that drain was a fifth of the window: 33.9 ms where 10 steps measure 28-29.5)
        res["pointnet2_config5"] = _leg_in_fresh_process(["--mode", "pointnet2", "--steps", "10", "--tiles", "16", "--points", "40000",
                                                          "--neighbors", "32"], timeout=400)

    leg("predict", "predict_config3", predict)
    leg("bf16", "bf16", bf16)
    leg("bf16x3", "bf16x3_split_products", bf16x3)
    leg("dropin", "dropin", dropin)
    leg("collective", "forced_collective_1rank", collective)
    leg("torch", "torch_rocm_baseline", torch_leg)
    if (N, K) == (12800, 16):
        leg("dense", "dense_tiles_config5", dense)
        leg("pointnet2", "pointnet2_config5", pointnet2)
    torch.cuda.empty_cache()


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
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,  # "nccl" IS RCCL on ROCm
                                device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: RCCL came up with {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from myria3d_amd import _lib

    _lib.lib()  # no fallback: fail here if the HIP library is missing
    if args.mode == "predict":
        res = predict_bench(args, dev, world, rank)
        if world == 1:  # (the whole chain from one resident cloud: a single-GPU measurement)
            res["end_to_end"] = predict_e2e_bench(args, dev)
        if rank == 0:
            print(json.dumps(res), flush=True)
    elif args.mode == "prepare":
        if world > 1:
            raise SystemExit("--mode prepare is a single-GPU run")
        prepare_bench(args, dev)
    elif args.mode == "dropin":
        if world > 1:
            raise SystemExit("--mode dropin is a single-GPU run")
        dropin_bench(args, dev)
    elif args.mode == "pointnet2":
        if world > 1:
            raise SystemExit("--mode pointnet2 is a single-GPU run")
        pointnet2_bench(args, dev)
    else:
        B, N, K = args.tiles, args.points, args.neighbors
        extras = world == 1 and not args.skip_extras and not args.force_collective
        res, net, pos, plan = train_bench(args, dev, world, rank, B, N, K, args.steps, args.warmup, with_eager=extras,
                                          precision=args.precision)
        if world > 1 or args.force_collective:
            res["rccl_ranks"] = dist.get_world_size()
        if rank == 0:
            if not args.skip_roofline:
                try:
                    _progress("rooflines")
                    rl = stage_rooflines(net, pos, plan)
                    res["roofline"] = rl["dominant"]
                    res["roofline_knn_lse_stage"] = rl["knn_lse"]
                except Exception as e:
                    res["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        del net, pos, plan
        if extras:
            _extra_legs(args, dev, res, B, N, K)
        if rank == 0:
            if world == 1 and not args.skip_cpu_baseline:
                _progress("CPU baseline")
                res["cpu_baseline"] = cpu_baseline(args.cpu_tiles, N, K, full=args.cpu_baseline_full)
            _progress("done")
            print(json.dumps(res), flush=True)
    if world > 1 or args.force_collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
