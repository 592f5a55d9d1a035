"""The launch form of the hot path that ``bench.py`` times, as a product class: ``GraphedStep``.

One step of ``Model.training_step`` (``/root/reference/myria3d/models/model.py:105-120``: forward, criterion,
``backward``) plus the optimizer step — or one eval forward (``model.py:79``) — over batches of a FIXED tile layout
(``ptr``; the reference's ``fixed_num_points`` preparation gives every tile the same size,
``configs/datamodule/transforms/preparations/fixed_num_points.yaml:19-24``), launched as hipGraphs:

* ``B[k]`` = the step on input buffer set ``k`` (it consumes the position-only tables — kNN grids and tables, encoder
  moments, random decimation, decoder 1-NN tables — that were computed one step earlier into geometry slot ``k``);
* ``A[k]`` = the position-only work for the NEXT step (buffer set ``k ^ 1`` -> geometry slot ``k ^ 1``).

``B[k]`` and ``A[k]`` are replayed on two streams and overlap; across steps they alternate ``k``.  Ordering:
``B_i`` waits for ``A_{i-1}`` (its tables); ``A_i`` waits for ``B_{i-1}`` (the last reader of the slot it rewrites) and
for ``load_next`` (the positions it reads).  Inside ONE graph the executor submits the position-only branch first
and the feature chain starts ~0.8 ms late, hence two graphs.

Data pipeline: two input buffer sets; ``load_next(x, pos, y)`` fills the set of the step AFTER the coming one while the
coming one still reads its own (``load(x, pos, y)`` fills the coming one and rebuilds its tables eagerly — first batch,
or after a gap).  With N > 1 ranks the gradient all-reduce (RCCL, capturable) and the Adam launch are PART of ``B[k]``
since round 4 (``collective="captured"``, the default): run outside the graphs — after ``B``, beside ``A`` — they cost
0.44 ms of graph start / stop bubbles per 4.6 ms step on one rank (round 3's ``forced_collective_1rank`` leg), i.e. 9 % of
per-GPU throughput the moment ``world_size > 1``.  If the stack refuses to capture the collective, ``prepare()`` falls
back to that form (``collective="eager"`` asks for it) and says so in ``self.collective``.

``launch="eager"`` runs the same step kernel by kernel (lookahead interleaved between the blocks of the forward);
``lookahead=False`` builds the tables inside the step (one graph, one buffer set).  Results are the same in every
form (``tests/test_gpu_train.py::test_graphed_step_*``).
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor

from .randla import HipRandLANet, make_plan
from .train import FusedAdam, cross_entropy


class _BufferSet:
    def __init__(self, n: int, num_features: int, device, with_labels: bool):
        self.x = torch.zeros((n, num_features), dtype=torch.float32, device=device)
        self.pos = torch.zeros((n, 3), dtype=torch.float32, device=device)
        self.y = torch.zeros(n, dtype=torch.int64, device=device) if with_labels else None
        self.out: Optional[Tensor] = None  # loss (train) / logits (eval) of the last step on this set


class GraphedStep:
    """``mode="train"``: ``step()`` = forward + cross-entropy (``ignore_index``) + backward + ``optimizer.step()``;
    returns the loss (a device scalar).  ``mode="eval"``: ``step()`` = eval forward under ``no_grad``; returns the
    logits.  ``ptr``: the tile layout every batch shares (host list or tensor)."""

    def __init__(self, net: HipRandLANet, ptr, num_features: int, *, mode: str = "train",
                 optimizer: Optional[FusedAdam] = None, ignore_index: int = 65, lookahead: bool = True,
                 launch: str = "graph", lookahead_mode: Optional[str] = None, optimizer_in_graph: Optional[bool] = None,
                 warmup: int = 2, collective: str = "captured", tune_streams: int = 6, accumulate: int = 1):
        if lookahead_mode is None:
            # two graphs on two streams pay off when the step is longer than the position-only chain (training: 4.72 vs
            # 4.79 ms).  Rounds 2-5: the eval forward was shorter than that chain and waited for it every step (1.59 vs
            # 1.18 ms) -> one graph.  Round 6: with the K-NN / 1-NN queries of the four levels as one launch each the chain
            # ends at 40 % of the eval forward, and two graphs win there too (0.99 vs 1.015 ms; bf16 0.757 vs 0.810:
            # profiles/r06t_eval_lookahead_mode_ab.log)
            lookahead_mode = "dual"
        if mode not in ("train", "eval") or launch not in ("graph", "eager") or lookahead_mode not in ("dual", "single"):
            raise ValueError("mode: train|eval, launch: graph|eager, lookahead_mode: dual|single")
        if mode == "train" and optimizer is None:
            raise ValueError("GraphedStep(mode='train') needs the FusedAdam that owns the net's flat buffers")
        # accumulate = k (Lightning's ``accumulate_grad_batches``; the reference's production run: 3,
        # configs/experiment/RandLaNet_base_run_FR.yaml:18): ``step()`` is a MICRO-batch — forward, loss, backward into the
        # flat gradient buffer, which the backward kernels add to — and every k-th call also runs the optimizer (one
        # all-reduce with N > 1 ranks, like DDP's no_sync, and an update with the mean gradient).  The optimizer then sits
        # outside the captured step (a graph is the same work at every replay).
        self.accumulate = int(accumulate)
        if self.accumulate < 1 or (self.accumulate > 1 and mode != "train"):
            raise ValueError("accumulate: a positive number of micro-batches per optimizer step (train mode)")
        if self.accumulate > 1:
            if optimizer_in_graph:
                raise ValueError("accumulate > 1 with optimizer_in_graph=True: the optimizer runs every k-th step only")
            optimizer_in_graph = False
        self._micro = 0
        self.net, self.opt, self.mode = net, optimizer, mode
        self.ignore_index = ignore_index
        self.lookahead, self.launch, self.lookahead_mode = bool(lookahead), launch, lookahead_mode
        dev = next(net.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedStep runs on an MI355X (cuda/HIP device) only")
        host_ptr = [int(v) for v in (ptr.tolist() if torch.is_tensor(ptr) else ptr)]
        self.ptr = torch.tensor(host_ptr, dtype=torch.int64, device=dev)
        self.plan = make_plan(host_ptr, net.decimation, net.num_neighbors, dev)
        n = host_ptr[-1]
        self.sets: List[_BufferSet] = [_BufferSet(n, num_features, dev, mode == "train")
                                       for _ in range(2 if self.lookahead else 1)]
        if collective not in ("captured", "eager"):
            raise ValueError("collective: captured|eager")
        multi = mode == "train" and optimizer.uses_collective()
        # with a gradient exchange the all-reduce is captured together with Adam (RCCL collectives are stream-ordered
        # kernels; torch's process group enqueues them on the capturing stream's side) unless collective="eager"
        self.collective = ("captured" if collective == "captured" else "eager") if multi else "none"
        self.opt_in_graph = (not multi or collective == "captured") if optimizer_in_graph is None else bool(optimizer_in_graph)
        if multi and self.opt_in_graph and collective == "eager":
            raise ValueError("optimizer_in_graph=True with collective='eager': the optimizer step contains the all-reduce")
        if multi and not self.opt_in_graph:
            self.collective = "eager"
        self.turn = 0
        self._tune_streams = int(tune_streams)
        self._warmup = max(1, warmup)
        self._graphs = None  # (gB, gA) once captured
        self._sA: Optional[torch.cuda.Stream] = None
        self._evA = self._evReady = None
        self._primed = False
        self.side_stream_ms: Optional[List[float]] = None  # ms per step measured for each candidate replay stream of graph A

    # ------------------------------------------------------------------------------------------
    def _cur(self) -> _BufferSet:
        return self.sets[self.turn % len(self.sets)]

    def load(self, x: Tensor, pos: Tensor, y: Optional[Tensor] = None) -> None:
        """Fill the buffer set of the COMING step and rebuild its position-only tables."""
        self._sync_geometry_stream()
        self._fill(self._cur(), x, pos, y)
        self._primed = False

    def load_next(self, x: Tensor, pos: Tensor, y: Optional[Tensor] = None) -> None:
        """Fill the buffer set of the step AFTER the coming one (the coming step prefetches its tables)."""
        if not self.lookahead:
            raise RuntimeError("load_next() needs lookahead=True (two buffer sets)")
        self._fill(self.sets[(self.turn + 1) & 1], x, pos, y)

    def load_all(self, x: Tensor, pos: Tensor, y: Optional[Tensor] = None) -> None:
        """The same batch into every buffer set (benchmarks: a static batch)."""
        self._sync_geometry_stream()
        for s in self.sets:
            self._fill(s, x, pos, y)
        self._primed = False

    def _sync_geometry_stream(self) -> None:
        """The previous step's graph ``A`` (other stream) reads the coming step's positions and writes its slot."""
        if self._sA is not None:
            torch.cuda.current_stream().wait_stream(self._sA)

    def _fill(self, s: _BufferSet, x, pos, y) -> None:
        s.x.copy_(x, non_blocking=True)
        s.pos.copy_(pos, non_blocking=True)
        if s.y is not None:
            s.y.copy_(y, non_blocking=True)
        if self._evReady is not None:
            self._evReady.record(torch.cuda.current_stream())

    # ------------------------------------------------------------------------------------------
    def _set_mode(self) -> None:
        want = self.mode == "train"
        if self.net.training != want:
            self.net.train(want)  # (walks 266 modules: only when it changes anything)

    def _body(self, k: int, prefetch: Optional[int], with_opt: bool):
        """The step on buffer set ``k``.  ``prefetch``: buffer set whose tables are enqueued, stage by stage, between the
        blocks of this forward (eager lookahead); None: no prefetch here."""
        net, s = self.net, self.sets[k]
        if self.mode == "train":
            if prefetch is not None:
                net.prefetch_geometry(self.sets[prefetch].pos, self.ptr, self.plan, train=True, interleave=True,
                                      slot=prefetch, owner=id(self))
            out = net(s.x, s.pos, None, self.ptr, plan=self.plan)
            loss = cross_entropy(out, s.y, ignore_index=self.ignore_index)  # model.py:118
            # backward nodes on THIS thread: one device per process leaves the engine's device thread nothing to overlap, and the
            # hand-off costs ~10 us for each of the ~190 nodes (2 ms -> 0.4 ms of host time per step: tools/host_profile.py st)
            with torch.autograd.set_multithreading_enabled(False):
                loss.backward(gradient=self._unit(loss))  # (a persistent 1: no ones_like fill node per step)
            if net.grad_side is not None:
                net.grad_side.join()  # (inside a capture the deferred leaf launches must be part of it)
            if self.lookahead:
                net.join_geometry()
            s.out = loss.detach()
            if with_opt:
                self.opt.step()
        else:
            with torch.no_grad():
                if prefetch is not None:
                    net.prefetch_geometry(self.sets[prefetch].pos, self.ptr, self.plan, train=False, interleave=True,
                                          slot=prefetch, owner=id(self))
                s.out = net(s.x, s.pos, None, self.ptr, plan=self.plan)
                if self.lookahead:
                    net.join_geometry()

    def _unit(self, like: Tensor) -> Tensor:
        u = getattr(self, "_one", None)
        if u is None or u.device != like.device or u.dtype != like.dtype:
            u = self._one = torch.ones((), dtype=like.dtype, device=like.device)
        return u

    def _geo(self, k: int) -> None:
        """The position-only work for buffer set ``k`` as a unit of its own (graph ``A``; also the eager priming)."""
        with torch.no_grad():
            self.net.prefetch_geometry(self.sets[k].pos, self.ptr, self.plan, train=self.mode == "train", slot=k,
                                       owner=id(self))
            self.net.join_geometry()

    def _state(self):
        net = self.net
        keep = [t for t in (net.flat_parameters, net.flat_grads) if t is not None] if net.flat_parameters is not None \
            else [p.data for p in net.parameters()]
        keep += [b for b in net.buffers()]
        if getattr(net, "_nbt_flat", None) is not None:
            keep.append(net._nbt_flat)  # (its last slot, the dropout step counter, is no registered buffer)
        if self.opt is not None:
            keep += [self.opt.exp_avg, self.opt.exp_avg_sq, self.opt.step_count]
        return keep

    def prepare(self, preserve_state: bool = True) -> "GraphedStep":
        """Warm up (allocator, zero arena, geometry slots of both buffer sets) and capture.  The warm-up runs real
        steps on whatever the buffer sets hold; ``preserve_state`` puts parameters, optimizer moments, BatchNorm
        statistics and the decimation RNG state back afterwards."""
        self._set_mode()
        net = self.net
        saved = [t.clone() for t in self._state()] if preserve_state else None
        nsets = len(self.sets)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # (warm-up off the default stream, as torch's graph recipe asks)
            for _ in range(self._warmup):
                for k in range(nsets):
                    if self.lookahead:
                        self._geo(k)
                    self._body(k, None, self.mode == "train")
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.launch == "graph":
            try:
                self._capture()
            except Exception as exc:  # noqa: BLE001 — any capture failure of the collective form falls back to the eager one
                if self.collective != "captured":
                    raise
                import warnings

                warnings.warn(f"GraphedStep: capturing the gradient all-reduce failed ({type(exc).__name__}: {exc}); "
                              "the all-reduce and the optimizer step run after the graph instead")
                torch.cuda.synchronize()
                self.collective, self.opt_in_graph = "eager", False
                self._capture()
        if self._tune_streams > 1 and self.lookahead and (
                (self._graphs is not None and self._graphs[1]) or (self._graphs is None and self.mode == "train")):
            self._pick_side_stream()
        if saved is not None:
            with torch.no_grad():
                for t, v in zip(self._state(), saved):
                    t.copy_(v)
            net.invalidate_eval_cache()
        self._primed = False
        torch.cuda.synchronize()
        return self

    def _pick_side_stream(self) -> None:
        """Which stream should graph ``A`` be replayed on?  HIP maps streams onto a few hardware queues (4 by default), and two
        streams that share one execute in enqueue order however independent their work is.  Which queue a new stream lands
        on depends on every stream created before it in the process — with an RCCL communicator alive (it owns streams of its
        own) the stream ``torch.cuda.Stream()`` hands out next sat on the step's queue, and the two graphs of a step ran one
        after the other: 4.45 + 0.69 = 5.14 ms instead of 4.60 (``tools/collective_probe.py``: the slowdown appears with the
        process group ALONE, no collective in the step — that, not the all-reduce, was round 3's "+0.44 ms of the N > 1 form").
        So the choice is measured: a few steps on each of ``tune_streams`` candidate streams, the fastest one is kept."""
        import time

        eager = self._graphs is None  # eager launching: the net's side stream carries the interleaved position-only work
        dev = self.ptr.device
        first = self.net._side_stream(dev) if eager else self._sA
        cands = [first] + [torch.cuda.Stream() for _ in range(self._tune_streams - 1)]
        ms = []
        for c in cands:
            if eager:
                self.net._finish_interleaved()
                self.net._look_queue.clear()
                torch.cuda.synchronize()
                self.net._streams[dev] = c
            else:
                self._sA = c
            self._primed = False
            for _ in range(2):
                self.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                self.step()
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) / 4 * 1e3)
        best = min(range(len(cands)), key=lambda i: ms[i])
        self.side_stream_ms = [round(v, 3) for v in ms]
        if eager:
            self.net._finish_interleaved()
            self.net._look_queue.clear()
            torch.cuda.synchronize()
            self.net._streams[dev] = cands[best]
            return
        self._sA = cands[best]
        self._evA.record()
        self._evReady.record()

    def _capture(self) -> None:
        net = self.net
        net._finish_interleaved()
        net._look_queue.clear()
        gB, gA = [], []
        with_opt = self.mode == "train" and self.opt_in_graph
        if self.lookahead:
            self._geo(0)  # pending tables for the first captured step
            torch.cuda.synchronize()
        if self.lookahead and self.lookahead_mode == "dual":
            for k in range(2):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):  # (RCCL's watchdog thread must not void it)
                    self._body(k, None, with_opt)
                gB.append(g)
                # captured WITH THE NET'S SIDE STREAM AS THE ORIGIN: the position-only work is enqueued on that
                # stream, so the graph is one chain.  Captured from another stream it is a fork / join around an
                # origin that holds no node of its own — and a replay of that form let the next step start on
                # half-written tables (tests/test_gpu_train.py::test_graphed_step_matches_plain_eager_steps)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=net._side_stream(self.ptr.device), capture_error_mode="thread_local"):
                    self._geo(k ^ 1)
                gA.append(g)
            self._sA = torch.cuda.Stream()
        elif self.lookahead:  # one graph per buffer set holding both branches
            for k in range(2):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._body(k, k ^ 1, with_opt)
                gB.append(g)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._body(0, None, with_opt)
            gB.append(g)
        net._finish_interleaved()
        net._look_queue.clear()  # (consumed inside the captures: nothing is pending for eager callers)
        self._graphs = (gB, gA)
        self._evA, self._evReady = torch.cuda.Event(), torch.cuda.Event()
        self._evA.record()
        self._evReady.record()

    def prime(self) -> None:
        """Eagerly build the position-only tables of the coming step (first step, after ``load()``, after a seed
        change).  Without lookahead there is nothing to prime."""
        if self.lookahead:
            k = self.turn & 1
            self._sync_geometry_stream()
            self.net._finish_interleaved()
            self.net._look_queue.clear()
            self._geo(k)
            if self._graphs is not None:
                self.net._look_queue.clear()  # the captured step reads slot k directly
                self._evA.record(torch.cuda.current_stream())
        self._primed = True

    # ------------------------------------------------------------------------------------------
    def step(self) -> Tensor:
        """One step; returns the loss (train) / logits (eval) tensor of this step's buffer set.  LIFETIME: the tensor lives in
        the captured graph's memory and is overwritten by the next step on the same buffer set (two steps later with
        lookahead, the very next step without) — read it (``.item()``) or ``.clone()`` it before then."""
        self._set_mode()
        if self._graphs is None and self.launch == "graph":
            self.prepare()
        if not self._primed:
            self.prime()
        k = self.turn % len(self.sets)
        cur = torch.cuda.current_stream()
        if self._graphs is not None:
            gB, gA = self._graphs
            if gA:
                cur.wait_event(self._evA)             # tables of this step (A of the previous step, or prime())
                self._sA.wait_event(self._evReady)    # previous step done with the slot A rewrites; next positions loaded
                gB[k].replay()
                with torch.cuda.stream(self._sA):
                    gA[k].replay()
                    self._evA.record(self._sA)
                self._evReady.record(cur)
            else:
                gB[k].replay()
        else:
            self._body(k, (k ^ 1) if self.lookahead else None, self.mode == "train" and self.opt_in_graph)
        if self.mode == "train" and not self.opt_in_graph:
            self._micro += 1
            if self._micro >= self.accumulate:
                # RCCL all-reduce + Adam: outside the graph, beside A on the other stream
                self.opt.step(grad_scale=1.0 / self.accumulate)
                self._micro = 0
        self.turn += 1
        return self.sets[k].out

    __call__ = step

    # ------------------------------------------------------------------------------------------
    def consumed_geometry(self, k: Optional[int] = None):
        """The position-only tables buffer set ``k`` (default: the last step's) was processed with — tests compare
        them with an eager run: ``(decimation indices per level [reference rows], level-1 kNN table [reference order])``.
        Only meaningful with lookahead (they live in the net's geometry slot ``k``) and before the next-but-one step."""
        from .randla import _knn_to_reference_order

        if not self.lookahead:
            raise RuntimeError("consumed_geometry() reads the lookahead slots")
        k = (self.turn - 1) & 1 if k is None else k
        geo = self.net._look_slots[(id(self), self.mode == "train", k)].geo
        return [d.clone() for d in geo.dec_ref], _knn_to_reference_order(geo.knn[0], geo.index[0])
