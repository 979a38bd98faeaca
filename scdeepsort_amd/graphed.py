"""hipGraph replay of the forward for repeated inference on one resident graph.

Small inputs (the reference's demo tissues, BASELINE cfg1 / cfg2) are launch-bound: a 2-layer forward is ~25 kernel
launches of a few microseconds each.  ``GraphedForward`` captures them once into a HIP graph (``torch.cuda.CUDAGraph``;
the C ABI only enqueues on the stream it is given, never synchronises or allocates, so it is capturable) and replays
it per call; features are copied into static buffers, logits come back in a static buffer.
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
