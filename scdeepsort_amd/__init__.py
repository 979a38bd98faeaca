"""scdeepsort_amd - MI355X-native hot path of scDeepSort (weighted GraphSAGE aggregation).

Public surface mirrors the reference: ``GNN`` (models/gnn.py:28), ``DeepSortPredictor`` /
``DeepSortClassifier`` (docs/api.rst:6-130).  The arithmetic runs in ``libwgnn_hip.so``
(C ABI in ``include/wgnn.h``); importing this package does not load it, using any
operator does, and fails loudly when it is missing.
"""
from ._lib import WgnnError, DST_IS_GENE, NO_ALPHA, SRC_IS_GENE
from .graph import AggCsr, CellGeneGraph, Plan, build_plan
from .gnn import GNN, NodeUpdate
from .api import DeepSortClassifier, DeepSortPredictor
from .graphed import GraphedForward, GraphedShardedForward, GraphedTrainStep
from .ops import agg_bwd_alpha, agg_bwd_src, agg_fwd, cross_entropy_sum, linear_fwd, weighted_mean_aggregate
from .sampler import DeviceSampler

__all__ = ["GNN", "NodeUpdate", "DeepSortClassifier", "DeepSortPredictor", "CellGeneGraph", "AggCsr", "Plan", "build_plan", "agg_fwd", "agg_bwd_src",
           "agg_bwd_alpha", "weighted_mean_aggregate", "linear_fwd", "DeviceSampler", "GraphedForward", "GraphedShardedForward", "GraphedTrainStep", "cross_entropy_sum",
           "WgnnError", "SRC_IS_GENE", "DST_IS_GENE", "NO_ALPHA"]
__version__ = "0.4.0"
