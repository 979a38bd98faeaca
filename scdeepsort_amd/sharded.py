"""Binding of the cell-shard orchestration (``dist.py``) to the HIP operators.

``ShardedWgnn`` owns one rank's shard of the graph: its cells' rows of the cells<-genes CSR,
the genes<-cells CSR restricted to its cells (globally normalised), and a replicated gene table.
With ``world_size == 1`` it degenerates to the plain single-GPU ``GNN.forward``.
"""
from __future__ import annotations

import contextlib

from typing import Optional

import torch
import torch.nn.functional as F

from . import dist as D
from ._lib import DST_IS_GENE, NO_ALPHA, SRC_IS_GENE
from .gnn import GNN, _is_relu, pad_width
from .graph import AggCsr, CellGeneGraph, build_plan
from .ops import agg_fwd, cross_entropy_sum, linear as _linear, linear_act, weighted_mean_aggregate, weighted_sum


class ShardedWgnn:
    def __init__(self, model: GNN, graph: CellGeneGraph, world: int, shard_sizes=None, pad_nnz: Optional[int] = None,
                 seed: int = 0):
        self.model, self.graph, self.world = model, graph, world
        # A cells<-genes pass that runs while the [G, H] all-reduce of the gene partial sums is in flight (dist.sharded_forward,
        # no-grad path) plans its ONE-round tile geometry for the CUs the communicator's workgroups leave (dist.COMM_CUS);
        # every other pass - the last layer, training (synchronous collectives), the gene side (it only ever meets the short
        # logits all-gather of the previous step, and its best 224-CU geometry costs +15 %) - keeps the whole chip.
        self.overlap_cu_budget = max(64, 256 - D.COMM_CUS) if world > 1 else 256
        self.shard_sizes = shard_sizes          # cells per rank (exchanged once at build): sync-free logits concat
        # the zero-padding of narrow hidden widths must not depend on the LOCAL shard size: the [G, Hp] partial sums and
        # the flat gradient bucket are all-reduced, so every rank must carry the same Hp.  ``pad_nnz`` = max over ranks.
        self.pad_nnz = graph.cg.nnz if pad_nnz is None else int(pad_nnz)
        for l in model.layers:
            if l.norm is not None or not (l.activation is None or _is_relu(l.activation)):
                raise ValueError("the sharded path fuses ReLU into the aggregation epilogue: norm / non-ReLU "
                                 "activations are not supported here (the reference never passes them, train.py:26-32)")
        self.relu = all(l.activation is not None for l in model.layers)
        if not self.relu and any(l.activation is not None for l in model.layers):
            raise ValueError("mixed per-layer activations are not supported on the sharded path")
        # train-mode dropout (gnn.py:33-36,60-64): replicated gene rows need the SAME mask on every rank -> one stream
        # seeded identically everywhere; every rank's own cells draw from a rank-local stream
        rank = D.world()[0]
        dev = graph.device
        self._gen_shared = torch.Generator(device=dev).manual_seed(1_000_003 * (seed + 1))
        self._gen_local = torch.Generator(device=dev).manual_seed(1_000_003 * (seed + 1) + 7919 * (rank + 1))

    @property
    def nnz(self) -> int:
        return self.graph.cg.nnz

    @staticmethod
    def gene_stats(col: torch.Tensor, raw: torch.Tensor, num_genes: int):
        """Per-gene (in-degree, raw weight sum) contributed by THIS shard's cells."""
        deg = torch.bincount(col.long(), minlength=num_genes).float()
        ssum = torch.zeros(num_genes, dtype=torch.float64, device=col.device)
        ssum.index_add_(0, col.long(), raw.double())
        return deg, ssum

    @staticmethod
    def build(model: GNN, rowptr: torch.Tensor, col: torch.Tensor, raw: torch.Tensor, num_genes: int,
              chunk: Optional[int] = None, global_stats=None, seed: int = 0) -> "ShardedWgnn":
        """``rowptr/col/raw``: device CSR of THIS rank's (cells x genes) raw expression.
        ``global_stats`` = (deg, sum) over ALL shards; when None and a process group is up they are all-reduced."""
        rank, world = D.world()
        g = CellGeneGraph.from_device_csr(rowptr, col, raw, num_genes, chunk)
        if world > 1 or global_stats is not None or D.comm_active():
            # gene side: w = deg_glob * x / sum_glob over ALL ranks' cells (preprocess_internal.py:17-23)
            gc = g.gc
            deg_loc, sum_loc = ShardedWgnn.gene_stats(col, raw, num_genes)
            if global_stats is None:
                g_deg = D.all_reduce_sum_(deg_loc.clone())
                g_sum = D.all_reduce_sum_(sum_loc.clone())
            else:
                g_deg, g_sum = global_stats
            row_of = torch.repeat_interleave(torch.arange(num_genes, device=col.device, dtype=torch.int32),
                                             (gc.rowptr[1:] - gc.rowptr[:-1]).long())
            gc.val, gc.inv_deg = D.rescale_gene_side(deg_loc, sum_loc, g_deg, g_sum, gc.val, row_of)
            gc._t = None
            gc._tile_plan = None
            world = max(world, 2)
        sizes, pad_nnz = None, None
        if D.comm_active():
            import torch.distributed as tdist
            mine = torch.tensor([g.num_cells, g.cg.nnz], dtype=torch.long, device=col.device)
            every = [torch.zeros_like(mine) for _ in range(D.world()[1])]
            tdist.all_gather(every, mine)
            sizes = [int(t[0].item()) for t in every]
            pad_nnz = max(int(t[1].item()) for t in every)          # rank-invariant width decision (see __init__)
        eng = ShardedWgnn(model, g, world, sizes, pad_nnz, seed)
        if D.comm_active():                                         # one-time check: equal carried widths on every rank
            import torch.distributed as tdist
            w = torch.tensor([W.shape[0] for W, _ in eng._weights()[:-1]], dtype=torch.long, device=col.device)
            every = [torch.zeros_like(w) for _ in range(D.world()[1])]
            tdist.all_gather(every, w)
            if any(not torch.equal(e, w) for e in every):
                raise RuntimeError(f"ranks disagree on the carried hidden widths: {[e.tolist() for e in every]}")
        return eng

    # -- local arithmetic bound to the HIP kernels (differentiable: K1 forward, K2/K3 backward) ---------
    def _ops(self) -> D.LocalOps:
        m, g = self.model, self.graph
        G = g.num_genes

        def cells_layer(p_g, p_c, b, relu, rows=None, self_compact=False):
            if rows is None:
                return weighted_mean_aggregate(g.cg, m.alpha, SRC_IS_GENE, G + 1, p_g, p_c, bias=b, relu=relu)
            return weighted_mean_aggregate(g.cg, m.alpha, SRC_IS_GENE, G + 1, p_g, p_c if self_compact else p_c[rows.long()],
                                           bias=b, relu=relu, row_ids=rows.to(torch.int32), self_compact=True)

        def cells_mean_linear(h_g, h_c, W, b, relu, rows=None, self_compact=False, prescaled=False):
            pre = h_g if prescaled else None              # h_g = alpha-folded rows from genes_finish(scale_out=True)
            if rows is None:
                z = weighted_mean_aggregate(g.cg, m.alpha, SRC_IS_GENE, G + 1, h_g, h_c, src_scaled=pre)
            else:
                z = weighted_mean_aggregate(g.cg, m.alpha, SRC_IS_GENE, G + 1, h_g, h_c if self_compact else h_c[rows.long()],
                                            row_ids=rows.to(torch.int32), self_compact=True, src_scaled=pre)
            return linear_act(z, W, b, relu)

        def fold_alpha_ok(width, n_seed_rows):
            from . import ops as _o                       # the f32 tile route is the only consumer of a folded table
            f32 = all(l.fc_neigh.weight.dtype == torch.float32 for l in m.layers)
            return f32 and _o.will_run_tiled(g.cg, width, n_seed_rows)

        def genes_partial(p_c):
            if torch.is_grad_enabled() and p_c.requires_grad:
                return weighted_sum(g.gc, p_c)
            return agg_fwd(g.gc, None, NO_ALPHA, 0, p_c, None, no_mean=True)

        def genes_finish(part, p_g, b, relu, scale_out=False):
            a = m.alpha.reshape(-1)
            if not (torch.is_grad_enabled() and (part.requires_grad or p_g.requires_grad or a.requires_grad)):
                # one K1 launch over an identity CSR: (alpha[r]*1*part[r] + alpha[G]*p_g[r]) * inv_deg[r] + b, ReLU fused;
                # scale_out: the finished row times alpha[r] once more = the next layer's alpha-folded source table
                return agg_fwd(self._identity(), a, DST_IS_GENE, G, part, p_g, bias=b, relu=relu, out_scale_alpha=scale_out)
            z = (a[:G].unsqueeze(1) * part + a[G] * p_g) * g.gc.inv_deg.unsqueeze(1) + b
            z = F.relu(z) if relu else z
            return z * a[:G].unsqueeze(1) if scale_out else z

        @contextlib.contextmanager
        def overlapped():
            saved, g.cg.cu_budget = g.cg.cu_budget, self.overlap_cu_budget
            try:
                yield
            finally:
                g.cg.cu_budget = saved

        return D.LocalOps(cells_layer, genes_partial, genes_finish, cells_mean_linear, fold_alpha_ok,
                          overlapped if self.overlap_cu_budget != 256 else None)

    def _identity(self) -> AggCsr:
        """G x G identity CSR carrying the GLOBAL gene-side 1/(deg+1): lets K1's epilogue finish the all-reduced sums."""
        if getattr(self, "_eye", None) is None:
            G, dev = self.graph.num_genes, self.graph.device
            rp = torch.arange(G + 1, dtype=torch.int32, device=dev)
            host = rp.cpu().numpy()
            self._eye = AggCsr(rp, torch.arange(G, dtype=torch.int32, device=dev), torch.ones(G, device=dev),
                               self.graph.gc.inv_deg, G, G, build_plan(host, device=dev), host)
        return self._eye

    def _weights(self):
        """(W, b) per layer + head.  On graphs that run the LDS-streamed kernels a hidden width below 256 is carried
        zero-padded to 256 columns (``GNN._pad_width``): pad rows / bias of every layer, input columns of the next."""
        m = self.model
        out, width_in = [], None
        for l in m.layers:
            W, b = l.fc_neigh.weight, l.fc_neigh.bias
            if width_in is not None and width_in > W.shape[1]:
                W = F.pad(W, (0, width_in - W.shape[1]))
            Hp = pad_width(self.pad_nnz, W.shape[0])
            if Hp != W.shape[0]:
                W, b = F.pad(W, (0, 0, 0, Hp - W.shape[0])), F.pad(b, (0, Hp - b.shape[0]))
            out.append((W, b))
            width_in = W.shape[0]
        Wo = m.linear.weight
        if width_in is not None and width_in > Wo.shape[1]:
            Wo = F.pad(Wo, (0, width_in - Wo.shape[1]))
        return out + [(Wo, m.linear.bias)]

    def dropout_masks(self, feats_g, feats_c_local):
        """Per-layer (gene mask, local cell mask) of one train-mode forward, or None (eval / dropout = 0)."""
        m = self.model
        if m.dropout is None or not m.training:
            return None
        p, dev = float(m.dropout.p), feats_g.device
        widths = [feats_g.shape[1]] + [W.shape[0] for W, _ in self._weights()[:-1]][: m.n_layers - 1]
        return [(D.dropout_mask((feats_g.shape[0], w), p, self._gen_shared, dev),
                 D.dropout_mask((feats_c_local.shape[0], w), p, self._gen_local, dev)) for w in widths]

    def train_step(self, feats_g, feats_c_local, labels_local, optimizer, seeds_local=None, dropout_masks=None,
                   sync_loss: bool = True):
        """Data-parallel full-batch step (cfg4): local CE-sum loss, SUM all-reduce of gradients, identical Adam step.
        Dropout (train.py:26-32 passes ``dropout`` into GNN) is applied to every layer's input rows like gnn.py:60-64.
        Returns the global loss as a float, or (``sync_loss=False``) as a 0-d device tensor - no host synchronisation per step."""
        self.model.train()
        if dropout_masks is None:
            dropout_masks = self.dropout_masks(feats_g, feats_c_local)
        return D.sharded_train_step(list(self.model.parameters()), self._weights, feats_g, feats_c_local, labels_local,
                                    self._ops(), self.model.n_layers, optimizer, seeds_local, dropout_masks, self.relu, _linear,
                                    cross_entropy_sum, sync_loss)

    def forward(self, feats_g: torch.Tensor, feats_c_local: torch.Tensor, gather_logits: bool = True,
                async_gather: bool = False) -> torch.Tensor:
        """Logits of every cell of the job (``gather_logits``) or of this rank's cells.  ``async_gather``: the concat of
        this call is left in flight; ``wait_gather()`` completes it before the returned tensor may be read.  The NEXT forward
        calls it by itself - after issuing its first layer's projections, before its first aggregation launch: the concat
        runs under the library GEMMs and never next to a tile pass (whose one-round geometry needs every CU it planned for)."""
        m = self.model
        if self.world == 1:
            return m.linear(m.embed(self.graph, (feats_g, feats_c_local)))
        if gather_logits and D.comm_active() and (self.shard_sizes is None or len(self.shard_sizes) != D.world()[1]):
            # an engine constructed directly (not through ``build``, which exchanges the sizes): one exchange, cached -
            # the per-forward concat then needs no size exchange / host read (``dist.sharded_forward`` raises without them)
            import torch.distributed as tdist
            mine = torch.tensor([self.graph.num_cells], dtype=torch.long, device=self.graph.device)
            every = [torch.zeros_like(mine) for _ in range(D.world()[1])]
            tdist.all_gather(every, mine)
            self.shard_sizes = [int(t.item()) for t in every]
        res = D.sharded_forward(self._weights(), None, feats_g, feats_c_local, self._ops(), m.n_layers, gather_logits,
                                self.shard_sizes, async_gather, self.dropout_masks(feats_g, feats_c_local), self.relu, _linear,
                                pre_aggregate=self.wait_gather)
        if async_gather:
            res, self._pending = res
        return res

    def wait_gather(self) -> None:
        work, self._pending = getattr(self, "_pending", None), None
        if work is not None:
            work.wait()

    def forward_alg_bytes(self, dense_dim: int, s0: int = 4) -> int:
        """Algorithmic HBM bytes of one 2-layer forward on this rank (SURVEY.md section 8d formula); ``s0`` = bytes per
        element of the layer-0 features (2 for fp16 storage), deeper layers are fp32."""
        g, m = self.graph, self.model
        G, C = g.num_genes, g.num_cells
        H = m.layers[0].fc_neigh.weight.shape[0]
        def b(nnz, R, S, din, dout, s):
            return 8 * nnz + 4 * (R + 1) + s * din * (S + R) + 4 * G + 4 * dout * R
        tot, din, s = 0, dense_dim, s0
        for i in range(m.n_layers):
            last = i == m.n_layers - 1
            if not last:
                tot += b(g.gc.nnz, G, C, din, H, s)
            tot += b(g.cg.nnz, C, G, din, H, s)
            din, s = H, 4
        return tot
