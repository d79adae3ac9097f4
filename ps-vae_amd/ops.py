"""Differentiable operators of the hot path: torch.autograd.Function wrappers whose forward
and backward are calls into the HIP library (ps_vae_amd.hip).  PyTorch only owns the tensors,
the autograd graph and the stream."""
import torch
from torch.autograd import Function

from . import hip

# Tests set this to a list: every fused Linear+ReLU output is appended in execution order
# (used to count ReLU outputs whose sign differs from a float64 evaluation).
RELU_TRACE = None

# Whether the fused encoder node carries the ReLU decisions of the messages as bits (left behind by
# the aggregation forward, read by its backward instead of the input rows).
import os as _os
MESSAGE_BITS = _os.environ.get('PSVAE_MESSAGE_BITS', '1') != '0'


# ------------------------------------------------------------------------ index containers
class Csr:
    """Adjacency for message passing.  (row_ptr, col, cls) is the CSR over DESTINATION nodes
    (col = source of each incoming edge, in edge order); (t_row_ptr, t_col, t_cls) the CSR
    over SOURCE nodes, used by the backward pass.  For a symmetric graph whose edge list is
    sorted by source — what collation produces — the two coincide."""

    def __init__(self, row_ptr, col, cls, t_row_ptr=None, t_col=None, t_cls=None):
        self.row_ptr, self.col, self.cls = row_ptr, col, cls
        self.t_row_ptr = row_ptr if t_row_ptr is None else t_row_ptr
        self.t_col = col if t_col is None else t_col
        self.t_cls = cls if t_cls is None else t_cls

    @property
    def num_nodes(self):
        return self.row_ptr.numel() - 1

    @classmethod
    def from_edge_index(cls, edge_index, edge_attr, num_nodes):
        """From the reference's tensors: edge_index [2,E] int64 (row 0 = source, row 1 =
        destination, src/modules/encoder.py:58) and one-hot edge_attr [E,C].  Index
        bookkeeping only (stable sorts keep the edge order inside every row)."""
        src, dst = edge_index[0], edge_index[1]
        if edge_attr.dim() == 2:
            klass = edge_attr.argmax(dim=1)
        else:
            klass = edge_attr
        klass = klass.to(torch.uint8)

        def build(key, other):
            order = torch.argsort(key, stable=True)
            counts = torch.bincount(key, minlength=num_nodes)
            ptr = torch.zeros(num_nodes + 1, dtype=torch.int32, device=key.device)
            ptr[1:] = torch.cumsum(counts, 0)
            return ptr, other[order].to(torch.int32).contiguous(), klass[order].contiguous()

        row_ptr, col, c = build(dst, src)
        t_row_ptr, t_col, t_c = build(src, dst)
        return cls(row_ptr, col, c, t_row_ptr, t_col, t_c)


class Segments:
    """Row ranges [begin[b], end[b]) of the graphs of a batch inside a node-major matrix."""

    def __init__(self, begin, end, covers_all_rows):
        self.begin, self.end, self.covers_all_rows = begin, end, covers_all_rows

    @property
    def num_graphs(self):
        return self.begin.numel()

    @classmethod
    def from_offsets(cls, node_off):
        return cls(node_off[:-1], node_off[1:], True)

    def counts(self):
        """Rows per graph, float32 [B] (made once)."""
        if getattr(self, '_counts', None) is None:
            self._counts = (self.end - self.begin).to(torch.float32)
        return self._counts

    def node_graph(self, rows):
        """Graph of every row, int32 [rows] (made once; segments that cover all `rows` rows only)."""
        if not self.covers_all_rows:
            raise ValueError('node_graph needs segments that cover all rows')
        if getattr(self, '_node_graph', None) is None:
            n = (self.end - self.begin).to(torch.int64)
            ids = torch.arange(self.num_graphs, dtype=torch.int32, device=self.begin.device)
            self._node_graph = torch.repeat_interleave(ids, n, output_size=rows)    # (the size given: no host stop)
        return self._node_graph


class PairList:
    """Candidate atom pairs in row-major (graph, m, n) order: node ids src/dst, graph id,
    ptr [B+1] (pairs of graph b are ptr[b]..ptr[b+1]) and the node range of every graph."""

    def __init__(self, src, dst, graph, ptr, nodes):
        self.src, self.dst, self.graph, self.ptr, self.nodes = src, dst, graph, ptr, nodes
        self.live = None   # device int32 [1] when the arrays are fixed-size buffers: only the first live[0] pairs count
        self.max_nodes = None   # nodes of the largest graph when the builder of the list knows it (sizes an LDS staging area)

    def __len__(self):
        return self.src.numel()

    @classmethod
    def from_edge_select(cls, edge_select):
        """edge_select [B,N,N] bool over PADDED node ids b*N + m
        (src/modules/vae_piece_decoder.py:99-109)."""
        B, N, _ = edge_select.shape
        b, m, n = edge_select.nonzero(as_tuple=True)
        dev = edge_select.device
        ptr = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        ptr[1:] = torch.cumsum(torch.bincount(b, minlength=B), 0)
        begin = (torch.arange(B, device=dev, dtype=torch.int32) * N)
        nodes = Segments(begin, begin + N, True)
        return cls((b * N + m).to(torch.int32), (b * N + n).to(torch.int32), b.to(torch.int32), ptr, nodes)


def _rows2d(t):
    """2-D float32 view with contiguous rows (copying only if the last dim is strided)."""
    if t.dim() != 2:
        t = t.reshape(-1, t.shape[-1])
    if t.dtype != torch.float32:
        t = t.float()
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:   # expanded rows
        t = t.contiguous()
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ------------------------------------------------------------------------ Linear
class _Linear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x2 = _rows2d(x)
        M, K = x2.shape
        N = weight.shape[0]
        w = weight if weight.stride(1) == 1 else weight.contiguous()
        out = torch.empty(M, N, dtype=torch.float32, device=x2.device)
        hip.gemm(M, N, K, x2, _ld(x2), 1, w, 1, w.stride(0), out, N, bias=bias, relu=relu)
        if relu and RELU_TRACE is not None:
            RELU_TRACE.append(out.detach())
        ctx.relu = relu
        ctx.has_bias = bias is not None
        ctx.lead = x.shape[:-1]
        ctx.save_for_backward(x2, w, out if relu else None)
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w, out = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        g2 = g.reshape(M, N)
        if g2.dtype != torch.float32 or not g2.is_contiguous():
            g2 = g2.float().contiguous()
        mask = out if ctx.relu else None
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, dtype=torch.float32, device=g2.device)
            hip.gemm(M, K, N, g2, N, 1, w, w.stride(0), 1, gx, K, amask=mask)
            gx = gx.view(*ctx.lead, K)
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if want_gb:
            gb = torch.empty(N, dtype=torch.float32, device=g2.device)
        if ctx.needs_input_grad[1]:
            # dW = dY^T X and, from the same pass over dY, db = dY^T 1
            gw = torch.empty(N, K, dtype=torch.float32, device=g2.device)
            hip.gemm(N, K, M, g2, 1, N, x2, _ld(x2), 1, gw, K, amask=mask, split_k=0,
                     rowsum=gb if want_gb else None)
        elif want_gb:
            hip.colsum(g2, gb, mask=mask)
        return gx, gw, gb, None


def linear(x, weight, bias=None, relu=False):
    """act(x W^T + b) — nn.Linear (+ ReLU) on the MFMA GEMM (src/modules/common_nn.py:15-20)."""
    return _Linear.apply(x, weight, bias, relu)


# ------------------------------------------------------------------------ GINE aggregate
class _GineAggregate(Function):
    @staticmethod
    def forward(ctx, x, table, csr):
        x2 = _rows2d(x)
        table = table.contiguous()
        out = torch.empty(x2.shape[0], x2.shape[1], dtype=torch.float32, device=x2.device)
        hip.gine_aggregate_fwd(x2, table, csr.row_ptr, csr.col, csr.cls, out)
        ctx.csr = csr
        ctx.save_for_backward(x2, table)
        return out

    @staticmethod
    def backward(ctx, g):
        x2, table = ctx.saved_tensors
        g2 = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        gx = torch.empty_like(g2)
        gt = torch.empty_like(table)
        csr = ctx.csr
        hip.gine_aggregate_bwd(x2, table, g2, csr.t_row_ptr, csr.t_col, csr.t_cls, gx, gt)
        return gx, gt, None


def gine_aggregate(x, table, csr):
    """x + sum over incoming edges of ReLU(x[source] + table[bond class])."""
    return _GineAggregate.apply(x, table, csr)


# ------------------------------------------------------------------------ fused GINE encoder
class _GineEncoder(Function):
    """Encoder.embed_node (src/modules/encoder.py:53-61) as ONE autograd node: node_trans,
    then t rounds of [aggregate -> Linear+ReLU -> Linear] with the SAME conv weights, every
    iterate written straight into its slot of the concatenated [V, t*H] output (no cat).

    Backward runs the layers in reverse with hand-placed kernels: the ReLU derivative is
    applied in the epilogue of the product that produces the gradient, the slot gradient of an iterate is added
    inside the aggregation backward, and the gradients of the shared conv weights are ONE
    product each over the t iterations stacked along K (K = t*V) with the bias gradient
    taken from the same pass.  ReLU decisions travel as bits: those of every layer's first Linear
    (one bit per element, left behind by the product's epilogue) and those of the messages (4 C bits
    per float4 column of a node, left behind by the aggregation), so the backward re-reads neither
    the hidden activations nor the aggregation's input rows.

    pool (Segments covering all rows): the node also takes embed_graph's sum over each graph's
    nodes (src/modules/encoder.py:63-67) and returns the pooled [B, t*H] instead of [V, t*H].
    Its backward then starts from ONE gradient row per graph and never spreads it to the nodes:
    the aggregation backward adds the graph's row where it writes a node's gradient; the last
    iterate, which feeds nothing but the pool, gets its input-gradient product once per graph
    (then spread and gated per node), and its weight-gradient block is a product over the graphs
    against the per-graph sums of the hidden rows."""

    @staticmethod
    def forward(ctx, x_in, w0, b0, table, w1, b1, w2, b2, csr, t, only_last=False, pool=None, track=True):
        """w0 None: x_in [V,H] already is node_trans' output (see embed_trans).
        only_last: return the last iterate alone (its slot of the internal buffer) - the caller that
        uses nothing else then gets a [V,H] gradient instead of a [V, t*H] one that is zero in t-1
        of its t slots."""
        x_in = _rows2d(x_in)
        V, Din = x_in.shape
        H = w1.shape[1]
        C = table.shape[0]
        dev = x_in.device
        w1, w2, table = w1.contiguous(), w2.contiguous(), table.contiguous()
        all_x = torch.empty(V, t * H, dtype=torch.float32, device=dev)
        a = torch.empty(t, V, H, dtype=torch.float32, device=dev)
        h = torch.empty(t, V, H, dtype=torch.float32, device=dev)
        if w0 is None:
            x0 = x_in if (x_in.dtype == torch.float32 and x_in.is_contiguous()) else x_in.float().contiguous()
            x_in = None
        else:
            w0 = w0.contiguous()
            x0 = torch.empty(V, H, dtype=torch.float32, device=dev)
            hip.gemm(V, H, Din, x_in, _ld(x_in), 1, w0, 1, Din, x0, H, bias=b0)
        # The ReLU decisions of every layer's first Linear as bits (V H / 8 bytes per layer): the
        # backward gates its gradient with them instead of re-reading h
        grad = track and any(ctx.needs_input_grad)    # (track: the caller's grad mode; inside forward it is always off)
        use_bits = V >= 32 and grad
        bits = torch.empty(t, hip.gemm_bits_words(V, H), dtype=torch.int64, device=dev) if use_bits else None
        # and those of the messages every node sends (H / 2 bytes per node and layer for C <= 4)
        use_masks = grad and C <= 8 and V > 0 and MESSAGE_BITS
        masks = None
        if use_masks:
            words, wtype = hip.gine_mask_words(V, H, C)
            masks = torch.empty(t, words, dtype=wtype, device=dev)
        prev = x0
        for i in range(t):
            if use_masks:
                hip.gine_aggregate_fwd_bits(prev, table, csr.row_ptr, csr.col, csr.cls, a[i], masks[i])
            else:
                hip.gine_aggregate_fwd(prev, table, csr.row_ptr, csr.col, csr.cls, a[i])
            if use_bits:
                hip.gemm_bits(V, H, H, a[i], H, 1, w1, 1, H, h[i], H, bias=b1, relu=True, relu_bits=bits[i])
            else:
                hip.gemm(V, H, H, a[i], H, 1, w1, 1, H, h[i], H, bias=b1, relu=True)
            if RELU_TRACE is not None:
                RELU_TRACE.append(h[i].detach())
            slot = all_x[:, i * H:(i + 1) * H]
            hip.gemm(V, H, H, h[i], H, 1, w2, 1, H, slot, t * H, bias=b2)
            prev = slot
        ctx.csr, ctx.t, ctx.only_last, ctx.pool = csr, t, only_last, pool
        ctx.shape = (V, H, C)
        # with the message bits the backward needs neither x0 nor the iterates
        ctx.save_for_backward(x_in, w0, table, w1, w2, None if use_masks else x0, None if use_masks else all_x, a, h, bits,
                              masks)
        if pool is not None:
            pooled = torch.empty(pool.num_graphs, t * H, dtype=torch.float32, device=dev)
            hip.segment_sum_fwd(all_x, pool.begin, pool.end, pooled)
            return pooled
        return all_x[:, (t - 1) * H:] if only_last else all_x

    @staticmethod
    def backward(ctx, g_all):
        x_in, w0, table, w1, w2, x0, all_x, a, h, bits, masks = ctx.saved_tensors
        csr, t, pool = ctx.csr, ctx.t, ctx.pool
        V, H, C = ctx.shape
        dev = a.device
        only_last = ctx.only_last      # then g_all is the gradient of the last iterate alone, [V,H]
        if g_all.dtype != torch.float32:
            g_all = g_all.float()
        if not g_all.is_contiguous():
            g_all = g_all.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        g_pool = None
        if pool is not None:
            if bits is not None:
                g_pool = g_all                         # [B, t*H]: one row per graph, kept that way
            else:                                      # small batches / wide class sets: spread it to the nodes
                spread = torch.empty(V, t * H, **f32)
                hip.segment_sum_bwd(g_all, pool.begin, pool.end, spread)
                g_all = spread
        # dL/d x_i (output of layer i) for i < t-1, stacked for the dW2 product; the gradient of the
        # last iterate is read where it arrives (a [V,H] block of g_all): no copy into the stack
        last = None if g_pool is not None else (g_all if only_last else g_all[:, (t - 1) * H:])
        gx = torch.empty(max(t - 1, 1), V, H, **f32)
        gh = torch.empty(t, V, H, **f32)      # dL/d (pre-activation of h_i): gated as it is produced
        ga = torch.empty(V, H, **f32)
        g_x0 = torch.empty(V, H, **f32)
        g_table = torch.empty_like(table)
        node_graph = pool.node_graph(V) if g_pool is not None else None

        # the t layers share the bond table: with the mask bits their table-gradient partials are kept as slabs
        # (one per workgroup and layer) and summed once behind the loop
        slabs = torch.empty(t, hip.gine_bwd_slabs(V, H, C), C, H, **f32) if masks is not None and V > 0 else None

        def aggregate_bwd(i, x_prev, out, add, seg_add, first):
            if masks is not None:
                hip.gine_aggregate_bwd_bits(masks[i], C, ga, csr.t_row_ptr, csr.t_col, csr.t_cls, out, g_table, add=add,
                                            seg_add=seg_add, node_graph=node_graph if seg_add is not None else None,
                                            accumulate_table=not first, slabs=None if slabs is None else slabs[i])
            else:
                hip.gine_aggregate_bwd(x_prev, table, ga, csr.t_row_ptr, csr.t_col, csr.t_cls, out, g_table, add=add,
                                       accumulate_table=not first, seg_add=seg_add,
                                       node_graph=node_graph if seg_add is not None else None)

        for i in range(t - 1, -1, -1):
            if g_pool is not None and i == t - 1:
                # the last iterate feeds only the pool: (g W2) once per graph, spread and gated per node
                q = torch.empty(g_pool.shape[0], H, **f32)
                g_last = g_pool[:, (t - 1) * H:]
                hip.gemm(g_pool.shape[0], H, H, g_last, _ld(g_last), 1, w2, H, 1, q, H)
                hip.graph_rows_gated(q, node_graph, bits[i], gh[i])
            else:
                g_i = last if i == t - 1 else gx[i]
                if bits is not None:                                            # through the last Linear + ReLU
                    hip.gemm_bits(V, H, H, g_i, _ld(g_i), 1, w2, H, 1, gh[i], H, gate_bits=bits[i])
                else:
                    hip.gemm(V, H, H, g_i, _ld(g_i), 1, w2, H, 1, gh[i], H, cmask=h[i])
            hip.gemm(V, H, H, gh[i], H, 1, w1, H, 1, ga, H)                     # through the first Linear
            first = i == t - 1
            if i > 0:
                slot = slice((i - 1) * H, i * H)
                add = None if (only_last or g_pool is not None) else g_all[:, slot]
                seg_add = g_pool[:, slot] if g_pool is not None else None
                aggregate_bwd(i, None if all_x is None else all_x[:, slot], gx[i - 1], add, seg_add, first)
            else:
                aggregate_bwd(i, x0, g_x0, None, None, first)
        if slabs is not None:
            hip.gine_table_reduce(slabs.view(-1, C, H), g_table)
        elif masks is not None:
            g_table.zero_()                      # (no node: no slab was written)
        need = ctx.needs_input_grad
        gw2, gb2 = torch.empty_like(w2), torch.empty(H, **f32)
        gw1, gb1 = torch.empty_like(w1), torch.empty(H, **f32)
        TV = t * V
        gh2, a2 = gh.view(TV, H), a.view(TV, H)
        if t > 1:     # layers 0 .. t-2 as one product over their stacked rows, then the last layer's block on top
            SV = (t - 1) * V
            hip.gemm(H, H, SV, gx.view(-1, H)[:SV], 1, H, h.view(TV, H)[:SV], H, 1, gw2, H, split_k=0, rowsum=gb2)
        if g_pool is not None:
            # sum over nodes of g[graph(v)]^T h[v] = sum over graphs of g[b]^T (sum of the graph's rows of h); with
            # the graph's row scaled by its node count and the row sums turned into means, the bias
            # gradient (the count-weighted sum of the graphs' rows) is the product's row sum
            B = g_pool.shape[0]
            hsum = torch.empty(B, H, **f32)
            hip.segment_sum_fwd(h[t - 1], pool.begin, pool.end, hsum)
            counts = pool.counts()
            hmean = hsum / counts.clamp(min=1.0)[:, None]
            gpn = g_pool[:, (t - 1) * H:] * counts[:, None]
            hip.gemm(H, H, B, gpn, 1, H, hmean, H, 1, gw2, H, split_k=0, rowsum=gb2, accumulate=t > 1)
        else:
            hip.gemm(H, H, V, last, 1, _ld(last), h[t - 1], H, 1, gw2, H, split_k=0, rowsum=gb2, accumulate=t > 1)
        hip.gemm(H, H, TV, gh2, 1, H, a2, H, 1, gw1, H, split_k=0, rowsum=gb1)
        tail = (None,) * 5
        if w0 is None:
            return (g_x0, None, None, g_table, gw1, gb1, gw2, gb2) + tail
        Din = x_in.shape[1]
        gw0, gb0 = torch.empty_like(w0), torch.empty(H, **f32)
        hip.gemm(H, Din, V, g_x0, 1, H, x_in, _ld(x_in), 1, gw0, Din, split_k=0, rowsum=gb0)
        g_in = None
        if need[0]:
            g_in = torch.empty(V, Din, **f32)
            hip.gemm(V, Din, H, g_x0, H, 1, w0, Din, 1, g_in, Din)
        return (g_in, gw0, gb0, g_table, gw1, gb1, gw2, gb2) + tail


def gine_encoder(x_in, w0, b0, table, w1, b1, w2, b2, csr, t, only_last=False, pool=None):
    """[V, t*H] concatenation of the t GINE iterates of node_trans(x_in); only_last: the last one, [V,H];
    pool (Segments that cover all rows): their sums over each graph's nodes, [B, t*H]."""
    if pool is not None and (only_last or not pool.covers_all_rows):
        raise ValueError('gine_encoder: pooling needs segments that cover all rows, and all iterates')
    return _GineEncoder.apply(x_in, w0, b0, table, w1, b1, w2, b2, csr, t, only_last, pool, torch.is_grad_enabled())


# ------------------------------------------------------------------------ embed_atom
class _EmbedAtom(Function):
    @staticmethod
    def forward(ctx, x, xp, xq, ea, ep, eq):
        ea, ep, eq = ea.contiguous(), ep.contiguous(), eq.contiguous()
        out = torch.empty(x.numel(), ea.shape[1] + ep.shape[1] + eq.shape[1], dtype=torch.float32, device=ea.device)
        hip.embed_atom_fwd(x, xp, xq, ea, ep, eq, out)
        ctx.save_for_backward(x, xp, xq)
        ctx.shapes = (ea.shape, ep.shape, eq.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x, xp, xq = ctx.saved_tensors
        g2 = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        (na, da), (npc, dp), (nq, dq) = ctx.shapes
        V, D = g2.shape
        rows = na + npc + nq
        if rows > hip.KEY_SORT_MAX_ROWS:
            # very large vocabularies: float atomics into zero-filled tables
            ga, gp, gq = (torch.zeros(s, dtype=torch.float32, device=g2.device) for s in ctx.shapes)
            hip.embed_atom_bwd(x, xp, xq, g2, ga, gp, gq)
            return None, None, None, ga, gp, gq
        # deterministic: the atoms in id order, then every table row = the sum of its run of rows of g
        perm, row_off, unit_off = hip.key_sort((x, xp, xq), (na, npc, nq))
        full = torch.empty(rows, D, dtype=torch.float32, device=g2.device)
        hip.keyed_row_sum(g2, perm, row_off, unit_off, full)
        return (None, None, None, full[:na, :da], full[na:na + npc, da:da + dp],
                full[na + npc:na + npc + nq, da + dp:])


def embed_atom(x, xp, xq, e_atom, e_piece, e_pos):
    """[E_atom[x] | E_piece[xp] | E_pos[xq]] for flat int32 id vectors -> [V, D]."""
    return _EmbedAtom.apply(x, xp, xq, e_atom, e_piece, e_pos)


# ------------------------------------------------------------------------ one embedding table
class _Embedding(Function):
    """table[ids] with a deterministic backward: the entries are put in id order by a stable
    counting sort and every table row is the sum of its run of rows of g in a fixed order
    (hip.key_sort + hip.keyed_row_sum), not atomic adds."""

    @staticmethod
    def forward(ctx, ids, table):
        table = table.contiguous()
        flat = ids.reshape(-1)
        out = torch.empty(flat.numel(), table.shape[1], dtype=torch.float32, device=table.device)
        hip.embedding_fwd(flat, table, out)
        ctx.save_for_backward(flat)
        ctx.rows = table.shape[0]
        return out.view(*ids.shape, table.shape[1])

    @staticmethod
    def backward(ctx, g):
        flat, = ctx.saved_tensors
        V, D = flat.numel(), g.shape[-1]
        g2 = g.reshape(V, D)
        g2 = g2 if (g2.dtype == torch.float32 and g2.is_contiguous()) else g2.float().contiguous()
        perm, row_off, unit_off = hip.key_sort((flat,), (ctx.rows,))
        full = torch.empty(ctx.rows, D, dtype=torch.float32, device=g2.device)
        hip.keyed_row_sum(g2, perm, row_off, unit_off, full)
        return None, full


def embedding(ids, table):
    """table[ids] for int32 ids of any shape -> [..., D] (D % 4 == 0)."""
    return _Embedding.apply(ids, table)


# ------------------------------------------------------------------------ first encoder layer on ids
class _StreamHandoff(Function):
    """Identity whose backward runs on the stream its forward was called on."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g


def stream_handoff(t):
    """t for use on another stream: gradients come back onto the stream current at this call."""
    return _StreamHandoff.apply(t) if t.requires_grad and torch.is_grad_enabled() else t


class EmbeddedIds:
    """The atoms of a batch as embedding ids plus the three tables, standing for the rows
    embed_atom would produce (src/modules/vae_piece_decoder.py:65-69).  Encoder.embed_node
    accepts it in place of those rows and computes node_trans from the ids (embed_trans)."""

    def __init__(self, x, xp, xq, e_atom, e_piece, e_pos):
        self.x, self.xp, self.xq = x, xp, xq
        self.tables = (e_atom, e_piece, e_pos)

    @property
    def shape(self):
        return (self.x.numel(), sum(t.shape[1] for t in self.tables))

    @property
    def device(self):
        return self.x.device

    def rows(self):
        """The embedded rows themselves, [V, D]."""
        return embed_atom(self.x, self.xp, self.xq, *self.tables)

    def for_side_stream(self):
        """The same ids with the block table passed through an identity node made HERE, on the current
        stream.  A chain that then uses the result on another stream sends its table gradient
        back through that node, i.e. onto this stream, where the tables' other users and their
        gradient accumulators live (autograd joins the two streams at the node)."""
        twin = EmbeddedIds(self.x, self.xp, self.xq, *self.tables)
        twin._block = stream_handoff(self.block_table())
        if getattr(self, '_sorted', None) is not None:
            twin._sorted = self._sorted
        return twin

    def tensors(self):
        """Every device buffer this object owns (for record_stream on a second stream)."""
        extra = [t for t in (getattr(self, '_sorted', None) or ()) + (getattr(self, '_block', None),) if t is not None]
        return (self.x, self.xp, self.xq) + tuple(extra)

    def block_table(self):
        """The three tables as ONE matrix [rows of all tables, D + 4]: table k sits in its own
        column block (zeros elsewhere) and column D holds a one on the atom rows (the bias of
        node_trans rides on it).  Built once per batch, on the stream current at the first call,
        and shared by every encoder."""
        if getattr(self, '_block', None) is None:
            self._block = _BlockTable.apply(*self.tables)
        return self._block

    def sorted_keys(self):
        """(perm, row_off, unit_off) of hip.key_sort over the three id vectors: the atoms in the
        order of the block table's rows, for the table gradient of every encoder.  Built once per
        batch; call it on the stream that created the ids BEFORE work forks to other streams: a
        backward on another stream only reads it."""
        if getattr(self, '_sorted', None) is None:
            self._sorted = hip.key_sort((self.x, self.xp, self.xq), [int(t.shape[0]) for t in self.tables])
        return self._sorted


class _EmbedTrans(Function):
    @staticmethod
    def forward(ctx, T, ids):
        T = T.contiguous()
        out = torch.empty(ids.x.numel(), T.shape[1], dtype=torch.float32, device=T.device)
        hip.embed_trans_fwd(ids.x, ids.xp, ids.xq, tuple(int(t.shape[0]) for t in ids.tables), T, out)
        ctx.ids, ctx.rows = ids, T.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        # dT[r] = sum of g over the atoms that looked row r up: a keyed row sum over the sorted atoms
        g2 = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        perm, row_off, unit_off = ctx.ids.sorted_keys()
        full = torch.empty(ctx.rows, g2.shape[1], dtype=torch.float32, device=g2.device)
        hip.keyed_row_sum(g2, perm, row_off, unit_off, full)
        return full, None


class _SplitColumns(Function):
    """Column blocks of a weight as views.  Autograd's own slicing would answer every block's gradient
    with a zero-filled copy of the whole weight plus an add per block; here the gradient of the
    weight is ONE concatenation of the blocks' gradients."""

    @staticmethod
    def forward(ctx, weight, *sizes):
        ctx.sizes, ctx.rows = sizes, weight.shape[0]
        out, col = [], 0
        for d in sizes:
            out.append(weight.narrow(1, col, d))
            col += d
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        ref = next(g for g in grads if g is not None)
        parts = [g if g is not None else ref.new_zeros(ctx.rows, d) for g, d in zip(grads, ctx.sizes)]
        return (torch.cat(parts, dim=1),) + (None,) * len(ctx.sizes)


def split_columns(weight, sizes):
    """weight[:, :s0], weight[:, s0:s0+s1], ... (the sizes must add up to the weight's width)."""
    if sum(sizes) != weight.shape[1]:
        raise ValueError(f'column blocks {tuple(sizes)} do not add up to the width {weight.shape[1]}')
    return _SplitColumns.apply(weight, *sizes)


class _BlockTable(Function):
    @staticmethod
    def forward(ctx, ea, ep, eq):
        (na, da), (npc, dp), (nq, dq) = ea.shape, ep.shape, eq.shape
        D = da + dp + dq
        Z = torch.empty(na + npc + nq, D + 4, dtype=torch.float32, device=ea.device)
        hip.block_table(ea.contiguous(), ep.contiguous(), eq.contiguous(), Z)   # one launch: blocks, zeros, ones column
        ctx.dims = (na, da, npc, dp, D)
        return Z

    @staticmethod
    def backward(ctx, g):
        na, da, npc, dp, D = ctx.dims
        return g[:na, :da], g[na:na + npc, da:da + dp], g[na + npc:, da + dp:D]


_ZERO_COLUMNS = {}


def _zero_columns(rows, device):
    key = (rows, device)
    if key not in _ZERO_COLUMNS:
        _ZERO_COLUMNS[key] = torch.zeros(rows, 3, dtype=torch.float32, device=device)
    return _ZERO_COLUMNS[key]


def embed_trans(ids, weight, bias):
    """node_trans(embed_atom(ids)) without the embedded rows (src/modules/encoder.py:54 over
    src/modules/vae_piece_decoder.py:65-69): T = block table x [weight | bias | 0]^T - one small
    product whose row r is table row r times the matching column block of `weight`, plus the bias
    on the atom rows (every atom has exactly one) - then T[x] + T[n_atom + xp] +
    T[n_atom + n_piece + xq] per atom."""
    widths = sum(int(tab.shape[1]) for tab in ids.tables)
    if widths != weight.shape[1]:
        raise ValueError(f'embedding widths sum to {widths}, node_trans expects {weight.shape[1]}')
    H = weight.shape[0]
    b = bias[:, None] if bias is not None else _zero_columns(H, weight.device)[:, :1]
    w_ext = torch.cat([weight, b, _zero_columns(H, weight.device)], dim=1)      # [H, D + 4]
    return _EmbedTrans.apply(linear(ids.block_table(), w_ext), ids)


# ------------------------------------------------------------------------ segment sum
class _SegmentSum(Function):
    @staticmethod
    def forward(ctx, x, segs):
        x2 = _rows2d(x)
        out = torch.empty(segs.num_graphs, x2.shape[1], dtype=torch.float32, device=x2.device)
        hip.segment_sum_fwd(x2, segs.begin, segs.end, out)
        ctx.segs, ctx.rows = segs, x2.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g2 = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        alloc = torch.empty if ctx.segs.covers_all_rows else torch.zeros
        gx = alloc(ctx.rows, g2.shape[1], dtype=torch.float32, device=g2.device)
        hip.segment_sum_bwd(g2, ctx.segs.begin, ctx.segs.end, gx)
        return gx, None


def segment_sum(x, segs):
    return _SegmentSum.apply(x, segs)


# ------------------------------------------------------------------------ rsample + KL
class _RsampleKl(Function):
    @staticmethod
    def forward(ctx, mean, lv_pre, eps):
        mean, lv_pre, eps = mean.contiguous(), lv_pre.contiguous(), eps.contiguous()
        z = torch.empty_like(mean)
        kl = torch.empty((), dtype=torch.float32, device=mean.device)
        hip.rsample_kl_fwd(mean, lv_pre, eps, z, kl)
        ctx.save_for_backward(mean, lv_pre, eps)
        return z, kl

    @staticmethod
    def backward(ctx, g_z, g_kl):
        mean, lv_pre, eps = ctx.saved_tensors
        g_z = None if g_z is None else g_z.contiguous()
        g_kl = None if g_kl is None else g_kl.contiguous()
        g_mean, g_lv = torch.empty_like(mean), torch.empty_like(mean)
        hip.rsample_kl_bwd(mean, lv_pre, eps, g_z, g_kl, g_mean, g_lv)
        return g_mean, g_lv, None


def rsample_kl(mean, lv_pre, eps):
    return _RsampleKl.apply(mean, lv_pre, eps)


# ------------------------------------------------------------------------ GRU (piece LM)
class _Gru(Function):
    """Single-layer GRU over a time-major sequence (the piece language model,
    src/modules/vae_piece_decoder.py:29,85-89): the input projection of ALL steps is one
    GEMM, each step is one GEMM for the hidden projection plus one fused gate kernel, and
    the weight gradients are one GEMM each over the steps stacked along K."""

    @staticmethod
    def forward(ctx, x, h0, w_ih, w_hh, b_ih, b_hh):
        T, B, E = x.shape
        H = h0.shape[1]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        x = x.contiguous()
        w_ih, w_hh = w_ih.contiguous(), w_hh.contiguous()
        gi = torch.empty(T, B, 3 * H, **f32)
        gates = torch.empty(T, B, 3 * H, **f32)
        hall = torch.empty(T + 1, B, H, **f32)
        if T * B:
            hip.gemm(T * B, 3 * H, E, x.view(T * B, E), E, 1, w_ih, 1, E, gi.view(T * B, 3 * H), 3 * H, bias=b_ih)
        ctx.seq = H % 4 == 0 and H <= hip.GRU_SEQ_MAX_H
        if ctx.seq:
            # the whole recurrence in one launch: a workgroup per 16 molecules walks all T steps
            gh = torch.empty(T, B, H, **f32)       # only the n-gate part is needed afterwards
            hip.gru_seq_fwd(gi, h0.contiguous(), w_hh, b_hh, hall, gates, gh)
        else:
            gh = torch.empty(T, B, 3 * H, **f32)
            hall[0].copy_(h0)
            for t in range(T):
                hip.gemm(B, 3 * H, H, hall[t], H, 1, w_hh, 1, H, gh[t], 3 * H, bias=b_hh)
                hip.gru_cell_fwd(gi[t], gh[t], hall[t], hall[t + 1], gates[t])
        ctx.save_for_backward(x, w_ih, w_hh, gh, gates, hall)
        ctx.has_bias = (b_ih is not None, b_hh is not None)
        return hall[1:]

    @staticmethod
    def backward(ctx, g_out):
        x, w_ih, w_hh, gh, gates, hall = ctx.saved_tensors
        T, B, E = x.shape
        H = hall.shape[2]
        f32 = dict(dtype=torch.float32, device=x.device)
        g_out = g_out if (g_out.dtype == torch.float32 and g_out.is_contiguous()) else g_out.float().contiguous()
        dgi = torch.empty(T, B, 3 * H, **f32)
        dgh = torch.empty(T, B, 3 * H, **f32)
        if ctx.seq:
            dh = torch.empty(B, H, **f32)
            hip.gru_seq_bwd(gates, gh, hall, g_out, w_hh.t().contiguous(), dgi, dgh, dh)
        else:
            dh = torch.zeros(B, H, **f32)          # gradient flowing into h_t from step t+1
            nxt = torch.empty(B, H, **f32)
            for t in range(T - 1, -1, -1):
                hip.gru_cell_bwd(gates[t], gh[t], hall[t], g_out[t], dh, dgi[t], dgh[t], nxt)
                hip.gemm(B, H, 3 * H, dgh[t], 3 * H, 1, w_hh, H, 1, nxt, H, accumulate=True)   # + dgh W_hh
                dh, nxt = nxt, dh
        g_wih, g_whh = torch.empty_like(w_ih), torch.empty_like(w_hh)
        g_bih = torch.empty(3 * H, **f32) if ctx.has_bias[0] else None
        g_bhh = torch.empty(3 * H, **f32) if ctx.has_bias[1] else None
        TB = T * B
        hip.gemm(3 * H, E, TB, dgi.view(TB, 3 * H), 1, 3 * H, x.view(TB, E), E, 1, g_wih, E, split_k=0, rowsum=g_bih)
        hip.gemm(3 * H, H, TB, dgh.view(TB, 3 * H), 1, 3 * H, hall[:T].reshape(TB, H), H, 1, g_whh, H, split_k=0,
                 rowsum=g_bhh)
        g_x = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty(T, B, E, **f32)
            hip.gemm(TB, E, 3 * H, dgi.view(TB, 3 * H), 3 * H, 1, w_ih, E, 1, g_x.view(TB, E), E)
        return g_x, dh, g_wih, g_whh, g_bih, g_bhh


def gru(x, h0, w_ih, w_hh, b_ih, b_hh):
    """x [T,B,E] time-major, h0 [B,H] -> hidden states [T,B,H]."""
    return _Gru.apply(x, h0, w_ih, w_hh, b_ih, b_hh)


# ------------------------------------------------------------------------ pair features
class _PairGather(Function):
    @staticmethod
    def forward(ctx, h, z, pairs):
        h2, z2 = _rows2d(h), _rows2d(z)
        feats = torch.empty(len(pairs), 2 * h2.shape[1] + z2.shape[1], dtype=torch.float32, device=h2.device)
        hip.pair_gather_fwd(h2, z2, pairs.src, pairs.dst, pairs.graph, feats)
        ctx.pairs, ctx.hshape, ctx.zshape = pairs, h2.shape, z2.shape
        return feats

    @staticmethod
    def backward(ctx, g):
        pairs = ctx.pairs
        g2 = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
        alloc = torch.empty if pairs.nodes.covers_all_rows else torch.zeros
        g_h = alloc(ctx.hshape, dtype=torch.float32, device=g2.device)
        g_z = torch.empty(ctx.zshape, dtype=torch.float32, device=g2.device)
        hip.pair_gather_bwd(g2 if len(pairs) else None, pairs.src, pairs.dst, pairs.ptr, pairs.nodes.begin,
                            pairs.nodes.end, g_h, g_z)
        return g_h, g_z, None


def pair_gather(h, z, pairs):
    """[h[src] | h[dst] | z[graph]] per candidate pair."""
    return _PairGather.apply(h, z, pairs)


class _PairCombine(Function):
    """act(ab[src][:F] + ab[dst][F:] + c[graph]) per candidate pair: the link predictor's first layer
    after its weight has been split by input block, so that the [P, 2H+L] feature rows are never
    formed - forward or backward."""

    @staticmethod
    def forward(ctx, ab, c, pairs, relu):
        ab2, c2 = _rows2d(ab), _rows2d(c)
        F = c2.shape[1]
        if ab2.shape[1] != 2 * F:
            raise ValueError(f'pair_combine: node rows of {ab2.shape[1]} columns, graph rows of {F}')
        out = torch.empty(len(pairs), F, dtype=torch.float32, device=ab2.device)
        max_nodes = getattr(pairs, 'max_nodes', None)
        if max_nodes and len(pairs) >= 4 * ab2.shape[0] and c2.shape[0] >= pairs.ptr.numel() - 1:
            # many more pairs than nodes (all candidate pairs of generated molecules): a graph's node rows go
            # through LDS once instead of through the caches once per pair
            hip.pair_combine_fwd_graphs(ab2[:, :F], ab2[:, F:], c2, pairs.src, pairs.dst, pairs.ptr, pairs.nodes.begin,
                                        pairs.nodes.end, max_nodes, out, relu=relu)
        else:
            hip.pair_combine_fwd(ab2[:, :F], ab2[:, F:], c2, pairs.src, pairs.dst, pairs.graph, out, relu=relu)
        if relu and RELU_TRACE is not None:
            RELU_TRACE.append(out.detach())
        ctx.pairs, ctx.relu, ctx.rows = pairs, relu, ab2.shape[0]
        ctx.save_for_backward(out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        pairs = ctx.pairs
        g2 = g if (g.dtype == torch.float32 and g.stride(-1) == 1) else g.float().contiguous()
        B, F = pairs.ptr.numel() - 1, g2.shape[1]
        alloc = torch.empty if pairs.nodes.covers_all_rows else torch.zeros
        g_ab = alloc(ctx.rows, 2 * F, dtype=torch.float32, device=g2.device)
        g_c = torch.empty(B, F, dtype=torch.float32, device=g2.device)
        hip.pair_combine_bwd(g2, out, ctx.relu, pairs.src, pairs.dst, pairs.ptr, pairs.nodes.begin, pairs.nodes.end,
                             g_ab, g_c)
        return g_ab, g_c, None, None


def pair_combine(ab, c, pairs, relu=True):
    """[P, F] rows act(ab[src[p]][:F] + ab[dst[p]][F:] + c[graph[p]]) for ab [V, 2F], c [B, F]."""
    return _PairCombine.apply(ab, c, pairs, relu)


# ------------------------------------------------------------------------ two Linears as one
class _AffineMerge(Function):
    @staticmethod
    def forward(ctx, w_last, b_last, w_out, b_out):
        w_last, w_out = w_last.contiguous(), w_out.contiguous()
        C, K = w_last.shape[0], w_out.shape[1]
        w_eff = torch.empty(C, K, dtype=torch.float32, device=w_last.device)
        b_eff = torch.empty(C, dtype=torch.float32, device=w_last.device)
        hip.affine_merge_fwd(w_last, b_last, w_out, b_out, w_eff, b_eff)
        ctx.save_for_backward(w_last, w_out, b_out)
        ctx.has_b_last = b_last is not None
        return w_eff, b_eff

    @staticmethod
    def backward(ctx, g_w_eff, g_b_eff):
        w_last, w_out, b_out = ctx.saved_tensors
        g_w_eff = g_w_eff.contiguous().float()
        g_b_eff = (torch.zeros(w_last.shape[0], dtype=torch.float32, device=w_last.device) if g_b_eff is None
                   else g_b_eff.contiguous().float())
        g_w_last, g_w_out = torch.empty_like(w_last), torch.empty_like(w_out)
        g_b_out = torch.empty_like(b_out) if b_out is not None else None
        hip.affine_merge_bwd(w_last, w_out, b_out, g_w_eff, g_b_eff, g_w_last, g_w_out, g_b_out)
        return g_w_last, (g_b_eff if ctx.has_b_last else None), g_w_out, g_b_out


def affine_merge(w_last, b_last, w_out, b_out):
    """(W, b) of x -> W_last (W_out x + b_out) + b_last: W = W_last W_out, b = W_last b_out + b_last."""
    return _AffineMerge.apply(w_last, b_last, w_out, b_out)


# ------------------------------------------------------------------------ link loss
class _LinkCe(Function):
    @staticmethod
    def forward(ctx, hid, weight, bias, golden, live=None):
        hid2 = _rows2d(hid)
        w = weight.contiguous()
        P, C = hid2.shape[0], w.shape[0]
        dev = w.device
        logits = torch.empty(P, C, dtype=torch.float32, device=dev)
        dlogits = torch.empty(P, C, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        n_correct = torch.empty((), dtype=torch.int32, device=dev)
        hip.link_ce_fwd(hid2, w, bias, golden, logits, dlogits, loss, n_correct, live=live)
        ctx.save_for_backward(hid2, w, dlogits)
        ctx.has_bias = bias is not None
        ctx.mark_non_differentiable(logits, n_correct)
        return loss, logits, n_correct

    @staticmethod
    def backward(ctx, g_loss, _g_logits, _g_ok):
        hid2, w, dlogits = ctx.saved_tensors
        P, F = hid2.shape
        C = w.shape[0]
        g_loss = g_loss.contiguous().float()
        g_hid = gw = gb = None
        if ctx.needs_input_grad[0]:
            g_hid = torch.empty(P, F, dtype=torch.float32, device=w.device)
            hip.link_ce_bwd_input(dlogits, w, g_loss, g_hid)
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        alloc = torch.empty if P else torch.zeros   # (both are written in full when there is a pair at all)
        if want_gb:
            gb = alloc(C, dtype=torch.float32, device=w.device)
        if ctx.needs_input_grad[1]:
            # (dW)^T = hid^T dlogits: F rows fill the 128-row panels, the few classes are columns
            gwt = alloc(F, C, dtype=torch.float32, device=w.device)
            if P:
                hip.gemm(F, C, P, hid2, 1, _ld(hid2), dlogits, C, 1, gwt, C, split_k=0)
            gw = gwt.t() * g_loss
        if want_gb and P:
            hip.colsum(dlogits, gb)
        if want_gb:
            gb = gb * g_loss
        return g_hid, gw, gb, None, None


def link_ce(hid, weight, bias, golden, live=None):
    """Last Linear of the link predictor + mean 4-class cross-entropy.
    Returns (loss, logits [P,C], n_correct).  live (device int32 [1]): only the first live[0] rows are
    pairs, the rest pads a fixed-size buffer and takes no part."""
    return _LinkCe.apply(hid, weight, bias, golden, live)


# ------------------------------------------------------------------------ the scalar tail of the step
class _MseSelected(Function):
    @staticmethod
    def forward(ctx, pred, props, sel):
        pred2 = pred if (pred.dtype == torch.float32 and pred.stride(-1) == 1) else pred.float().contiguous()
        props2 = props if (props.dtype == torch.float32 and props.stride(-1) == 1) else props.float().contiguous()
        diff = torch.empty(pred2.numel(), dtype=torch.float32, device=pred2.device)
        loss = torch.empty((), dtype=torch.float32, device=pred2.device)
        hip.mse_selected_fwd(pred2, props2, sel, diff, loss)
        ctx.save_for_backward(diff)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        diff, = ctx.saved_tensors
        g_pred = torch.empty_like(diff)
        hip.mse_selected_bwd(diff, g.contiguous().float(), g_pred)
        return g_pred.view(ctx.shape), None, None


def mse_selected(pred, props, sel):
    """nn.MSELoss()(pred, props[:, sel]) (src/pl_models/ps_vae_model.py:55-57) in one launch each way.
    pred [B,k], props [B,*] f32, sel int64 [k] on the device."""
    return _MseSelected.apply(pred, props, sel)


class _LossMix(Function):
    @staticmethod
    def forward(ctx, rec, pred, kl, alpha, beta):
        out = torch.empty((), dtype=torch.float32, device=rec.device)
        hip.loss_mix_fwd(rec.float(), pred.float(), kl.float(), alpha, beta, out)
        ctx.alpha, ctx.beta = alpha, beta
        return out

    @staticmethod
    def backward(ctx, g):
        g3 = torch.empty(3, dtype=torch.float32, device=g.device)
        hip.loss_mix_bwd(g.contiguous().float(), ctx.alpha, ctx.beta, g3)
        return g3[0], g3[1], g3[2], None, None


def loss_mix(rec, pred, kl, alpha, beta):
    """alpha rec + (1 - alpha) pred + beta kl on device scalars (src/pl_models/ps_vae_model.py:71-74), one launch
    each way; beta: a Python number or a device scalar (no gradient flows to it)."""
    return _LossMix.apply(rec, pred, kl, alpha, beta)


# ------------------------------------------------------------------------ token loss with an ignored label
class _TokenCe(Function):
    @staticmethod
    def forward(ctx, logits, gold, ignore_index, n_classes):
        x = _rows2d(logits)
        R, width = x.shape
        dev = x.device
        dlogits = torch.empty(R, width, dtype=torch.float32, device=dev)
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        counts = torch.empty(2, dtype=torch.int32, device=dev)
        hip.token_ce_fwd(x, gold, ignore_index, n_classes, dlogits, loss, counts)
        ctx.save_for_backward(dlogits, loss)
        ctx.shape = logits.shape
        ctx.mark_non_differentiable(counts)
        return loss[0], counts

    @staticmethod
    def backward(ctx, g_loss, _g_counts):
        dlogits, loss = ctx.saved_tensors
        # the saved rows are softmax - onehot: times g_loss / #tokens they are the gradient
        return (dlogits * (g_loss.float() * loss[1])).view(ctx.shape), None, None, None


def token_ce(logits, gold, ignore_index, n_classes=None):
    """Mean cross-entropy over the rows whose label is not `ignore_index`
    (nn.CrossEntropyLoss(ignore_index=...)).  logits [..., V] f32, gold int32 [...] (same leading
    shape, flattened).  n_classes < V: only the first n_classes columns are classes, the rest is
    padding (ignored, zero gradient).  Returns (loss, counts) with counts = int32 [#rows counted,
    #rows whose argmax is the label]."""
    width = logits.shape[-1]
    return _TokenCe.apply(logits.reshape(-1, width), gold.reshape(-1), ignore_index, width if n_classes is None else n_classes)


def linear_relu_classes(x, weight, bias, cls_weight, cls_bias):
    """relu(x W^T + b) C^T + c for a class layer C of at most 4 rows, without gradients and without the
    [M, N] intermediate in memory (psvae_gemm_f32_relu_classes: the class layer is applied per column tile in
    the product's epilogue, the tiles' parts are added here).  None when the shape is not covered."""
    M, K = x.shape
    N, C = weight.shape[0], cls_weight.shape[0]
    tiles = hip.relu_classes_tiles(N)
    if tiles == 0 or C > 4 or K % 4 or M == 0 or x.stride(1) != 1 or x.stride(0) % 4 or not weight.is_contiguous():
        return None
    w4 = cls_weight.detach()
    if C < 4:
        w4 = torch.cat([w4, w4.new_zeros(4 - C, N)], dim=0)
    partials = torch.empty(M, tiles, 4, dtype=torch.float32, device=x.device)
    hip.gemm_relu_classes(x, weight.detach(), None if bias is None else bias.detach(), w4.contiguous(), partials)
    logits = partials.sum(dim=1)[:, :C]
    return logits + cls_bias.detach() if cls_bias is not None else logits
