"""hipGraph replay of the forward for repeated inference on one resident graph.

Small inputs (the reference's demo tissues, BASELINE cfg1 / cfg2) are launch-bound: a 2-layer forward is ~25 kernel
launches of a few microseconds each.  ``GraphedForward`` captures them once into a HIP graph (``torch.cuda.CUDAGraph``;
the C ABI only enqueues on the stream it is given, never synchronises or allocates, so it is capturable) and replays
it per call; features are copied into static buffers, logits come back in a static buffer.

``GraphedTrainStep`` does the same for one mini-batch TRAINING step (train.py:71-87: forward on a seed batch, CE-sum
loss, backward, Adam): the seed sub-plan and the batch's source-major block are built on the device with static shapes
(``AggCsr.subplan`` / ``seed_block_transposed``), so the whole step is free of host synchronisation and replays as ONE
graph launch per batch - the reference pays a sampler call, a feature copy and a device->host sync per layer per batch
(gnn.py:50).
"""
from __future__ import annotations

from typing import Optional

import torch

from .gnn import GNN
from .graph import CellGeneGraph


class GraphedForward:
    def __init__(self, model: GNN, graph: CellGeneGraph, features: torch.Tensor, seeds: Optional[torch.Tensor] = None,
                 warmup: int = 2):
        if model.training:
            raise ValueError("capture an eval-mode model (dropout draws a fresh mask per call)")
        self.model, self.graph = model, graph
        self.features = features.clone()
        self.seeds = None if seeds is None else seeds.clone()
        side = torch.cuda.Stream(device=graph.device)
        side.wait_stream(torch.cuda.current_stream(graph.device))
        with torch.cuda.stream(side), torch.no_grad():          # warm-up outside capture: plans, LDS attributes, allocator
            for _ in range(max(1, warmup)):
                self.logits = model(graph, self.features, seeds=self.seeds)
        torch.cuda.current_stream(graph.device).wait_stream(side)
        with torch.cuda.device(graph.device):                   # the capture stream is opened on the CURRENT device
            self._g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g), torch.no_grad():
                self.logits = model(graph, self.features, seeds=self.seeds)

    def __call__(self, features: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Replays the captured forward (on new ``features`` of the same shape if given).  The returned tensor is the
        static output buffer: clone it to keep it across calls."""
        if features is not None:
            self.features.copy_(features)
        self._g.replay()
        return self.logits


class GraphedTrainStep:
    """``step_fn(batch_ids) -> loss`` (a 0-d device tensor) must do forward + backward + optimizer step with a
    capturable optimizer (``torch.optim.Adam(..., capturable=True)``) and no host synchronisation.  The first ``warmup``
    calls run eagerly (they are real training steps); the next full-size batch is captured and every later batch of the
    same size replays the graph.  Batches of another size (the tail of an epoch) run eagerly."""

    def __init__(self, step_fn, batch_size: int, device: torch.device, warmup: int = 3):
        self.step_fn, self.batch_size, self.device, self.warmup = step_fn, int(batch_size), device, warmup
        self._calls, self._g, self._ids, self._loss = 0, None, None, None
        self._side = torch.cuda.Stream(device=device)
        self.replays = 0

    def _eager(self, batch):
        cur = torch.cuda.current_stream(self.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            loss = self.step_fn(batch)
        cur.wait_stream(self._side)
        return loss

    def __call__(self, batch: torch.Tensor) -> torch.Tensor:
        if batch.shape[0] != self.batch_size:
            return self._eager(batch)
        self._calls += 1
        if self._calls <= self.warmup:
            return self._eager(batch)
        if self._g is None:
            self._ids = batch.clone()
            with torch.cuda.device(self.device):                 # capture on the device that owns the model, whatever is current
                self._g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._g):
                    self._loss = self.step_fn(self._ids)
        else:
            self._ids.copy_(batch)
        self._g.replay()
        self.replays += 1
        return self._loss


class GraphedShardedForward:
    """hipGraph replay of the SHARDED no-grad forward (``ShardedWgnn.forward`` at world > 1): ~15 launches and two collectives.
    Measured on one MI355X (round 4, ``profiles/r04_shard_trace.json``): a rank's shard of the cfg3 job is GPU-bound even at
    12.5k cells (N = 8), so the replay is NOT faster than eager issue there (0.637 vs 0.596 ms) - this class exists for
    smaller / launch-bound shards and for callers that want one launch per forward; ``bench.py`` uses it with ``--graphed on``.

    * ``nccl`` (RCCL): the collectives are captured WITH the kernels - the async [G, H] all-reduce on the communicator's
      stream (forked from / joined to the capture stream by events, so its overlap with the cells<-genes pass is part of the
      graph) and the logits all-gather - one graph launch per forward (``mode == "whole"``).
    * any other backend (gloo: collectives run on the host; the shared-GPU debug mode of ``bench.py`` and the tests): the
      capture is cut at every collective (``dist.COLLECTIVE_HOOK``): kernels replay as graph segments, the collectives in
      between are re-issued eagerly on the same static tensors (``mode == "segments"``).

    ``__call__`` returns the static output buffer (all cells' logits when ``gather_logits``), valid until the next call."""

    def __init__(self, engine, feats_g: torch.Tensor, feats_c_local: torch.Tensor, gather_logits: bool = True,
                 warmup: int = 2, mode: Optional[str] = None):
        import torch.distributed as tdist
        from . import dist as D
        if engine.model.training:
            raise ValueError("capture an eval-mode model (dropout draws a fresh mask per call)")
        if engine.world == 1:
            raise ValueError("world == 1: use GraphedForward")
        self.engine, self.gather = engine, gather_logits
        dev = engine.graph.device
        self.device = dev
        self.feats_g, self.feats_c = feats_g.clone(), feats_c_local.clone()
        if mode is None:
            mode = "whole" if (not D.comm_active() or tdist.get_backend() == "nccl") else "segments"
        self.mode = mode
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():          # warm-up: plans, LDS attributes, allocator, communicator set-up
            for _ in range(max(1, warmup)):
                self.logits = engine.forward(self.feats_g, self.feats_c, gather_logits=gather_logits)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._seq = []                                          # replay program: CUDAGraph objects and collective thunks
        with torch.cuda.device(dev):
            if mode == "whole":
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g), torch.no_grad():
                    self.logits = engine.forward(self.feats_g, self.feats_c, gather_logits=gather_logits)
                self._seq.append(g)
            else:
                self._capture_segments(side)
        self.n_graphs = sum(isinstance(x, torch.cuda.CUDAGraph) for x in self._seq)
        self.n_eager_collectives = len(self._seq) - self.n_graphs

    def _capture_segments(self, side):
        from . import dist as D
        pool = torch.cuda.graph_pool_handle()                   # one memory pool: later segments read earlier segments' tensors
        cur = {"g": None}

        def begin():
            cur["g"] = torch.cuda.CUDAGraph()
            cur["g"].capture_begin(pool=pool)

        def end():
            cur["g"].capture_end()
            self._seq.append(cur["g"])

        def hook(fn):                                           # a data-path collective: cut the graph here
            end()
            torch.cuda.current_stream(self.device).synchronize()
            fn()                                                # eager, synchronous (see dist.sharded_forward)
            self._seq.append(fn)
            begin()

        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            D.COLLECTIVE_HOOK = hook
            try:
                begin()
                self.logits = self.engine.forward(self.feats_g, self.feats_c, gather_logits=self.gather)
                end()
            except BaseException:
                if torch.cuda.is_current_stream_capturing():       # never leave the stream in capture mode behind an error
                    try:
                        cur["g"].capture_end()
                    except Exception:
                        pass
                raise
            finally:
                D.COLLECTIVE_HOOK = None
        torch.cuda.current_stream(self.device).wait_stream(side)

    def __call__(self, feats_g: Optional[torch.Tensor] = None, feats_c_local: Optional[torch.Tensor] = None) -> torch.Tensor:
        if feats_g is not None:
            self.feats_g.copy_(feats_g)
        if feats_c_local is not None:
            self.feats_c.copy_(feats_c_local)
        for x in self._seq:
            if isinstance(x, torch.cuda.CUDAGraph):
                x.replay()
            else:
                x()                                             # host-side collective on the segment's static tensors
        return self.logits
