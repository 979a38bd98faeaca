"""``GNN`` - drop-in counterpart of the reference's ``models/gnn.py:28-68``.

Same constructor signature, same parameter names / shapes (so released
``{'model': state_dict}`` checkpoints load, train.py:117-123 / predict.py:56-59):

    layers.{i}.fc_neigh.weight [H, D_in | H]   layers.{i}.fc_neigh.bias [H]
    alpha [gene_num + 2, 1]                    linear.weight [n_cls, H]   linear.bias [n_cls]

What changes is the operand of ``forward``: instead of a DGL ``NodeFlow`` built per
seed batch (train.py:71-81) it takes the HBM-resident :class:`CellGeneGraph` and
evaluates the layers over the whole graph at once: each seed's
logits depend only on its full L-hop in-neighbourhood, which the reference's
sampler takes in full (``expand_factor`` = all nodes, train.py:37-38), so the
result is identical per seed while the per-batch recomputation of the closure
disappears.  ``seeds`` selects / orders the output rows exactly like
``nf.layer_parent_nid(-1)`` (train.py:81, predict.py:75-76).

Per layer (gnn.py:60-65, 18-25) with ``Z = mean-aggregate`` and ``NodeUpdate = relu(W Z + b)``:
the aggregation is linear, so when ``H <= D_in`` the projection is applied first
(``P = h W^T`` once for all nodes, then ``relu(agg(P) + b)`` fused in the kernel
epilogue), which narrows every gathered row from D_in to H floats.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._lib import DST_IS_GENE, SRC_IS_GENE
from .graph import CellGeneGraph
from . import ops as _ops
from .ops import linear as _linear, linear_act as _linear_act, weighted_mean_aggregate, weighted_sum


class NodeUpdate(nn.Module):
    """Parameter holder with the reference's names (gnn.py:10-16); applied by GNN.forward."""

    def __init__(self, in_feats, out_feats, activation=None, norm=None):
        super().__init__()
        self.fc_neigh = nn.Linear(in_features=in_feats, out_features=out_feats)
        self.activation = activation
        self.norm = norm
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=nn.init.calculate_gain('relu'))


def pad_width(nnz: int, H: int) -> int:
    """Hidden width as the kernels carry it: the next multiple of 4 (the kernels move float4s; the extra columns get zero
    weights / bias, stay 0 through ReLU and meet zero weight columns in the next layer).  Round 1 carried EVERY width
    below 256 as 256 columns on large graphs because only the D = 256 tile kernel was hand-scheduled; since round 2 that
    kernel runs any D <= 256 natively (1 KiB LDS slots, D/4 lanes in the DMA), so e.g. the reference default
    hidden_dim = 200 (train.py:137) moves 200-float rows.  ``ops.PAD_NARROW_TO_256`` restores the old behaviour for
    A/B timing; ``nnz`` (max over ranks in a sharded job, so that every rank decides alike) only matters then."""
    from . import ops
    if ops.PAD_NARROW_TO_256 and H < 256 and ops.TILED_MIN_WORK is not None and nnz * max(H, 128) >= ops.TILED_MIN_WORK:
        return 256
    return -(-H // 4) * 4


def _is_relu(fn) -> bool:
    return fn in (F.relu, torch.relu) or isinstance(fn, nn.ReLU)


class GNN(nn.Module):
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, gene_num, activation=None, norm=None, dropout=0.0):
        super().__init__()
        self.n_layers = n_layers
        self.gene_num = gene_num
        self.dropout = nn.Dropout(p=dropout) if dropout != 0 else None
        self.layers = nn.ModuleList()
        self.layers.append(NodeUpdate(in_feats, n_hidden, activation=activation, norm=norm))
        for _ in range(n_layers - 1):
            self.layers.append(NodeUpdate(n_hidden, n_hidden, activation=activation, norm=norm))
        # [gene_num] is alpha of gene-gene (self-loop), [gene_num+1] of the cell self-loop (gnn.py:42-43)
        self.alpha = nn.Parameter(torch.ones(gene_num + 2, 1, dtype=torch.float32))
        self.linear = nn.Linear(n_hidden, n_classes)
        nn.init.xavier_uniform_(self.linear.weight, gain=nn.init.calculate_gain('relu'))
        self.order = "auto"          # "auto" | "project_first" | "aggregate_first"
        self.fold_alpha = True       # no-grad forward: the layer below the last writes its gene rows alpha-folded (A/B switch)

    # -- one NodeFlow block, both node types ------------------------------------------------------
    def _pad_width(self, g: CellGeneGraph, H: int) -> int:
        return pad_width(g.cg.nnz, H)

    def _project_first(self, layer: NodeUpdate, want_genes: bool) -> bool:
        # "auto": aggregate at the narrower width.  At EQUAL widths both orders aggregate the same bytes, but project-first
        # multiplies every source row too (genes AND cells), aggregate-first only the rows this layer outputs: the last layer
        # (cells only) then skips the [G, H] x [H, H] product - and is the reference's literal order (gnn.py:65-66).
        W = layer.fc_neigh.weight
        return self.order == "project_first" or (self.order == "auto" and (
            W.shape[0] < W.shape[1] or (W.shape[0] == W.shape[1] and want_genes)))

    def _layer(self, g: CellGeneGraph, layer: NodeUpdate, h_g: torch.Tensor, h_c: torch.Tensor,
               want_genes: bool, cell_rows: Optional[torch.Tensor], h_c_compact: bool = False,
               genes_out_scaled: bool = False, h_g_prescaled: bool = False):
        """One NodeFlow block for both node types.  ``cell_rows``: compute only these cell rows (a seed batch);
        ``h_c_compact``: ``h_c`` already holds one row per entry of ``cell_rows`` (the previous layer was restricted to
        the seeds) - then cells cannot be sources at this layer (``want_genes`` is False).
        ``genes_out_scaled`` (no-grad path, decided by ``embed``): the gene rows this layer outputs are written as
        ``alpha[g] * h_g'[g]`` by the aggregation epilogue; ``h_g_prescaled``: ``h_g`` IS such a table (this layer reads gene
        rows only as the source of its cells<-genes pass, which then skips its scale launch)."""
        G = self.gene_num
        if h_c_compact and (want_genes or cell_rows is None):
            raise ValueError("compact cell rows can only feed the seeds' own self-loop")
        W, b = layer.fc_neigh.weight, layer.fc_neigh.bias
        project_first = self._project_first(layer, want_genes)
        if (genes_out_scaled and not (project_first and want_genes)) or (h_g_prescaled and (project_first or want_genes)):
            raise ValueError("alpha-folded gene tables pass from a project-first layer to a cells-only aggregate-first layer")
        if h_g.shape[1] % 4:                               # e.g. dense_dim = 50: zero feature columns up to a multiple of 4
            extra = -h_g.shape[1] % 4
            h_g, h_c = F.pad(h_g, (0, extra)), F.pad(h_c, (0, extra))
        if h_g.shape[1] > W.shape[1]:                      # input carried padded (see _pad_width): zero weight columns
            W = F.pad(W, (0, h_g.shape[1] - W.shape[1]))
        Hp = self._pad_width(g, W.shape[0]) if project_first else W.shape[0]
        if Hp != W.shape[0] and layer.norm is None:
            W, b = F.pad(W, (0, 0, 0, Hp - W.shape[0])), F.pad(b, (0, Hp - b.shape[0]))
        act = layer.activation
        fuse_relu = _is_relu(act)
        if self.dropout is not None:                       # node rows, before the gather (gnn.py:62-64)
            h_g, h_c = self.dropout(h_g), self.dropout(h_c)
        # fp16-stored features (BASELINE cfg5) meet fp32 weights: fp16-rounded inputs, fp32 multiply-accumulate.  On the no-grad
        # project-first path `ops.linear` widens them in the GEMM's loader (wgnn_linear_fwd_ex) - no fp32 copy of [C, D_in] in
        # HBM; everywhere else they are converted here.
        if h_g.dtype != W.dtype and not (project_first and h_g.dtype == torch.float16 and _ops.use_wgnn_linear(h_c, W)):
            h_g, h_c = h_g.to(W.dtype), h_c.to(W.dtype)

        def finish(x):
            if act is not None and not fuse_relu:
                x = act(x)
            if layer.norm is not None:
                x = layer.norm(x)
            return x

        compact = cell_rows is not None
        if project_first:
            # when the cells<-genes pass will run LDS-streamed (it needs alpha[g] * P_g[g] as its source table) and nothing
            # here is differentiated, ONE kernel writes both P_g and its alpha-folded copy (no scale_rows launch)
            p_g_scaled = None
            tiled_cells = (_ops.tiled_kernel_serves(g.cg, Hp) and not torch.is_grad_enabled()
                           and (cell_rows is None or cell_rows.shape[0] >= _ops.SEED_FULL_PASS_MIN_FRAC * g.cg.n_rows))
            need_all_cells = (want_genes or not compact) and not h_c_compact
            p_c_all = None
            joint = self._adjacent_rows(h_g, h_c) if (need_all_cells and not torch.is_grad_enabled()) else None
            if tiled_cells and _ops.use_wgnn_linear(h_g, W, dual=True):
                p_g, p_g_scaled = _ops.linear_fwd(h_g, W, row_scale=self.alpha.reshape(-1)[:G])
            elif joint is not None and not _ops.use_wgnn_linear(h_c, W):
                # gene and cell rows are one contiguous [G + C, D] table (features given as one tensor, preprocess_internal.py:202)
                # and the graph is small: ONE projection GEMM instead of two latency-bound ones (cfg2: 17 vs 27 us)
                p_all = _linear(joint, W)
                p_g, p_c_all = p_all[:G], p_all[G:]
            else:
                p_g = _linear(h_g, W)
            if need_all_cells and p_c_all is None:
                p_c_all = _linear(h_c, W)
            self_compact = compact
            if h_c_compact:
                p_c_self = _linear(h_c, W)
            elif not compact:
                p_c_self = p_c_all
            elif p_c_all is not None:                    # every cell's projection exists (the next layer's genes read it):
                p_c_self, self_compact = p_c_all, False  # the seeds' self rows are read in place, no [B, H] gather
            else:
                p_c_self = _linear(h_c[cell_rows.long()], W)
            out_c = weighted_mean_aggregate(g.cg, self.alpha, SRC_IS_GENE, G + 1, p_g, p_c_self, bias=b,
                                            relu=fuse_relu, row_ids=cell_rows, self_compact=self_compact, src_scaled=p_g_scaled)
            out_g = None
            if want_genes:
                out_g = weighted_mean_aggregate(g.gc, self.alpha, DST_IS_GENE, G, p_c_all, p_g, bias=b, relu=fuse_relu,
                                                out_scale_alpha=genes_out_scaled)
            return (finish(out_g) if out_g is not None else None), finish(out_c)
        # aggregate first (the reference's literal order: neigh -> fc_neigh -> activation)
        hc_self = h_c if (not compact or h_c_compact) else h_c[cell_rows.long()]
        z_c = weighted_mean_aggregate(g.cg, self.alpha, SRC_IS_GENE, G + 1, h_g, hc_self, row_ids=cell_rows,
                                      self_compact=compact, src_scaled=h_g if h_g_prescaled else None)
        out_c = _linear_act(z_c, W, b, fuse_relu)
        out_g = None
        if want_genes:
            z_g = weighted_mean_aggregate(g.gc, self.alpha, DST_IS_GENE, G, h_c, h_g)
            out_g = finish(_linear_act(z_g, W, b, fuse_relu))
        return out_g, finish(out_c)

    JOINT_PROJECTION_MAX_ROWS = 32768     # above this the two projections are bandwidth- / flop-bound and have their own tuned picks

    def _adjacent_rows(self, h_g: torch.Tensor, h_c: torch.Tensor) -> Optional[torch.Tensor]:
        """``[h_g; h_c]`` as ONE tensor without a copy when the two are consecutive row ranges of the same contiguous table
        (``embed`` slices a single ``features`` tensor that way), for small graphs; else None."""
        if (h_g.dim() != 2 or h_c.dim() != 2 or h_g.dtype != h_c.dtype or h_g.shape[1] != h_c.shape[1]
                or h_g.shape[0] + h_c.shape[0] > self.JOINT_PROJECTION_MAX_ROWS
                or not (h_g.is_contiguous() and h_c.is_contiguous())
                or h_g.untyped_storage().data_ptr() != h_c.untyped_storage().data_ptr()
                or h_c.storage_offset() != h_g.storage_offset() + h_g.numel()):
            return None
        return torch.as_strided(h_g, (h_g.shape[0] + h_c.shape[0], h_g.shape[1]), (h_g.shape[1], 1), h_g.storage_offset())

    def _fold_alpha_into_gene_rows(self, g: CellGeneGraph, layer: NodeUpdate, cell_rows) -> bool:
        """Whether the second-to-last layer may write its gene rows alpha-folded (see ``embed``): that layer runs
        project-first with a fused (or no) activation and no norm, the last layer runs aggregate-first at a width the tile
        kernel carries unpadded, and its cells<-genes pass takes the LDS-streamed route (the one reading a pre-folded table)."""
        nxt = self.layers[-1]
        act_ok = layer.norm is None and (layer.activation is None or _is_relu(layer.activation))
        H = layer.fc_neigh.weight.shape[0]
        Hp = self._pad_width(g, H)
        f32 = layer.fc_neigh.weight.dtype == torch.float32 and nxt.fc_neigh.weight.dtype == torch.float32   # the consumer of a
        # folded table is the f32 tile route only (a .half() model would otherwise meet the row-wave guard)
        return (getattr(self, "fold_alpha", True) and f32 and act_ok and self._project_first(layer, True) and not self._project_first(nxt, False)
                and Hp == H and nxt.fc_neigh.weight.shape[1] == H
                and _ops.will_run_tiled(g.cg, H, None if cell_rows is None else int(cell_rows.shape[0])))

    def embed(self, g: CellGeneGraph, features: torch.Tensor, seeds: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Cell embeddings = ``nf.layers[-1].data['activation']`` (gnn.py:66) for ``seeds`` (node ids >= G)."""
        G = self.gene_num
        if isinstance(features, (tuple, list)):            # (gene rows, cell rows) kept as separate tensors
            h_g, h_c = features
            if h_g.shape[0] != G or h_c.shape[0] != g.num_cells:
                raise ValueError("features tuple must be ([G, D], [C, D])")
        else:
            if features.shape[0] != g.num_nodes:
                raise ValueError(f"features must have {g.num_nodes} rows (genes then cells, preprocess_internal.py:202)")
            h_g, h_c = features[:G], features[G:]
        cell_rows = None
        if isinstance(seeds, range):
            # a contiguous block of cells (predict.py:61-88: the test cells of a predict graph are nodes G+n_support .. N-1).
            # When it is a sizeable share of a tile-kernel operand the cheapest evaluation is the plain full pass, sliced:
            # no seed gather / scatter at all (the support cells' rows are computed alongside and dropped).
            if seeds.step != 1 or seeds.start < G or seeds.stop > G + g.num_cells:
                raise ValueError("a seed range must be a step-1 range of cell node ids")
            H0 = self.layers[0].fc_neigh.weight.shape[0]
            big = (_ops.tiled_kernel_serves(g.cg, -(-min(H0, 256) // 4) * 4)
                   and len(seeds) >= _ops.SEED_FULL_PASS_MIN_FRAC * g.num_cells)
            if big:
                return self.embed(g, features, None)[seeds.start - G: seeds.stop - G]
            seeds = torch.arange(seeds.start, seeds.stop, device=g.device)
        if seeds is not None:
            cell_rows = (seeds.to(g.device) - G).to(torch.int32)
        # Which cell rows a layer must produce: cells are never sources for cells, so below the last layer a cell's
        # activation is read only (a) by the genes of the NEXT layer - all cells, as long as that layer computes genes -
        # and (b) by the cell's own self-loop.  With a seed list, the second-to-last layer therefore needs the seeds'
        # rows only (the last layer computes no genes): a 2-layer forward on a seed batch runs ONE full pass (layer-1
        # genes) instead of two; the reference's NodeFlow closure contains exactly these nodes.
        compact = False
        prescaled = False
        for i, layer in enumerate(self.layers):
            last = i == self.n_layers - 1
            rows = cell_rows if (cell_rows is not None and i >= self.n_layers - 2) else None
            # The gene rows of the layer below a cells-only last layer are read ONCE more: as the source table of that layer's
            # cells<-genes pass, which wants them alpha-folded (gnn.py:54, (h*alpha)*w).  With nothing to differentiate and
            # that pass on the LDS-streamed kernel, this layer's epilogue writes them folded (no scale launch in between).
            emit = (i == self.n_layers - 2 and not torch.is_grad_enabled() and self._fold_alpha_into_gene_rows(g, layer, cell_rows))
            h_g, h_c = self._layer(g, layer, h_g, h_c, want_genes=not last, cell_rows=rows, h_c_compact=compact,
                                   genes_out_scaled=emit, h_g_prescaled=prescaled)
            prescaled = emit
            compact = rows is not None
        H = self.layers[-1].fc_neigh.weight.shape[0]
        return h_c if h_c.shape[1] == H else h_c[:, :H]

    def embed_sampled(self, g: CellGeneGraph, features, nodeflow) -> torch.Tensor:
        """Seed-cell embeddings over a drawn NodeFlow (``sampler.sample_nodeflow``): the ``num_neighbors > 0`` mode
        of train.py:37-40.  Aggregate-first (the reference's literal order) on the compact destination rows; the
        mean divides by the number of DRAWN edges, the self-loop counts only where it was drawn."""
        G = self.gene_num
        if isinstance(features, (tuple, list)):
            h_g, h_c = features
        else:
            h_g, h_c = features[:G], features[G:]
        a = self.alpha.reshape(-1)
        n_blocks = len(nodeflow.blocks)
        for i, (layer, (cb, gb)) in enumerate(zip(self.layers, nodeflow.blocks)):
            if self.dropout is not None:
                h_g, h_c = self.dropout(h_g), self.dropout(h_c)
            if h_g.shape[1] % 4:                           # the kernels move float4s: zero columns up to a multiple of 4
                extra = -h_g.shape[1] % 4
                h_g, h_c = F.pad(h_g, (0, extra)), F.pad(h_c, (0, extra))

            def update(z):
                W = layer.fc_neigh.weight
                if z.shape[1] > W.shape[1]:
                    W = F.pad(W, (0, z.shape[1] - W.shape[1]))
                x = F.linear(z, W, layer.fc_neigh.bias)
                if layer.activation is not None:
                    x = layer.activation(x)
                if layer.norm is not None:
                    x = layer.norm(x)
                return x

            z_c = weighted_mean_aggregate(cb.csr, self.alpha, SRC_IS_GENE, G + 1, h_g, None)
            z_c = z_c + (a[G + 1] * cb.self_drawn * cb.csr.inv_deg).unsqueeze(1) * h_c[cb.rows]
            out_c = update(z_c)
            if i == n_blocks - 1:
                return out_c
            nh_c = torch.zeros((h_c.shape[0], out_c.shape[1]), dtype=out_c.dtype, device=out_c.device)
            nh_c = nh_c.index_copy(0, cb.rows, out_c)
            nh_g = torch.zeros((G, out_c.shape[1]), dtype=out_c.dtype, device=out_c.device)
            if gb is not None:
                s_g = weighted_sum(gb.csr, h_c)
                z_g = (a[gb.rows].unsqueeze(1) * s_g + (a[G] * gb.self_drawn).unsqueeze(1) * h_g[gb.rows]) \
                    * gb.csr.inv_deg.unsqueeze(1)
                nh_g = nh_g.index_copy(0, gb.rows, update(z_g))
            h_g, h_c = nh_g, nh_c
        raise AssertionError("unreachable")

    def forward(self, g: CellGeneGraph, features: Optional[torch.Tensor] = None,
                seeds: Optional[torch.Tensor] = None, num_neighbors: int = 0,
                generator: Optional[torch.Generator] = None, nodeflow=None) -> torch.Tensor:
        """Logits ``[len(seeds), n_classes]`` (all cells in order when ``seeds`` is None); no softmax (gnn.py:66-68).
        ``seeds``: node ids (>= G) as a tensor in any order, or a Python ``range`` for a contiguous block of cells.

        ``num_neighbors > 0`` draws a NodeFlow with at most that many in-edges per node (train.py:37-40); ``generator`` is
        a ``sampler.DeviceSampler`` (K5 ``wgnn_sample_rows``: static shapes, sync-free, hipGraph-capturable) or a
        ``torch.Generator`` (torch-op sampler with data-dependent shapes); 0 / None = every in-edge, the reference's
        default and its eval / predict mode."""
        if features is None:
            features = getattr(g, "features", None)
            if features is None:
                raise ValueError("pass features or set graph.features")
        if nodeflow is None and num_neighbors:
            from .sampler import DeviceSampler, sample_nodeflow, sample_nodeflow_static
            if isinstance(seeds, range):
                seeds = torch.arange(seeds.start, seeds.stop, seeds.step, device=g.device)
            cells = (seeds.to(g.device) - self.gene_num) if seeds is not None \
                else torch.arange(g.num_cells, device=g.device)
            if isinstance(generator, DeviceSampler):        # K5: static shapes, no host synchronisation, capturable
                nodeflow = sample_nodeflow_static(g, cells, self.n_layers, int(num_neighbors), generator)
            else:
                nodeflow = sample_nodeflow(g, cells, self.n_layers, int(num_neighbors), generator)
        if nodeflow is not None:
            return self.linear(self.embed_sampled(g, features, nodeflow))
        return self.linear(self.embed(g, features, seeds))
