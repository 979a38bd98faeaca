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
