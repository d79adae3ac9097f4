"""GINE graph encoder with the reference's API and state_dict keys
(src/modules/encoder.py:39-80).

embed_node / embed_graph take either the reference's tensors (edge_index [2,E] int64 with
one-hot edge_attr; graph_ids with an optional node mask) or the index containers collation
produces (ops.Csr, ops.Segments), in which case no index preparation runs at all."""
import torch
import torch.nn as nn

from .. import ops
from .common_nn import MLP, run_linear


class GINEConv(nn.Module):
    """Parameter holder with torch_geometric.nn.GINEConv's state_dict layout (`nn.*` and the
    buffer `eps` [1], as constructed at src/modules/encoder.py:48).  eps is fixed at 0
    (train_eps=False upstream), so out = nn(x_i + sum_j ReLU(x_j + e_ji))."""

    def __init__(self, nn_module):
        super(GINEConv, self).__init__()
        self.nn = nn_module
        self.register_buffer('eps', torch.zeros(1))

    def forward(self, x, edge_index, edge_attr):
        """edge_index: ops.Csr; edge_attr: the [C,H] table of transformed bond classes."""
        return self.nn(ops.gine_aggregate(x, edge_attr, edge_index))


_IDENTITY = {}


def _identity(n, device):
    key = (n, device)
    if key not in _IDENTITY:
        _IDENTITY[key] = torch.eye(n, dtype=torch.float32, device=device)
    return _IDENTITY[key]


class Encoder(nn.Module):
    def __init__(self, dim_in, num_edge_type, dim_hidden, dim_out, t=4):
        super(Encoder, self).__init__()
        self.num_edge_type = num_edge_type
        self.t = t
        self.node_trans = nn.Linear(dim_in, dim_hidden)
        self.edge_trans = nn.Linear(num_edge_type, dim_hidden)
        self.conv = GINEConv(MLP(dim_hidden, dim_hidden, dim_hidden, nn.ReLU, 2))
        self.linear = nn.Linear(dim_hidden * self.t, dim_out)

    def edge_table(self):
        """edge_trans applied to the C one-hot rows: row c = W[:, c] + b (the reference
        multiplies the one-hot edge_attr [E,C] by W^T, src/modules/encoder.py:55)."""
        if self.edge_trans.weight.is_cuda:
            # as the product it is in the reference: identity rows through the Linear (one launch forward, one
            # backward; the sum w + 0 + ... + 0 + b rounds as w + b does)
            return ops.linear(_identity(self.num_edge_type, self.edge_trans.weight.device), self.edge_trans.weight,
                              self.edge_trans.bias)
        return self.edge_trans.weight.t() + self.edge_trans.bias

    def embed_node(self, x, edge_index, edge_attr=None, want_all=True):
        """(last iterate [V,H], concatenation of the t iterates [V, t*H]) as src/modules/encoder.py:53-61.
        want_all=False (a caller that only uses the last iterate, as the decoder does,
        src/modules/vae_piece_decoder.py:93-96) returns (last, None) and spares the backward pass a
        [V, t*H] gradient that is zero outside its last slot."""
        if isinstance(edge_index, ops.Csr):
            csr = edge_index
        else:
            if edge_attr.dim() != 2 or edge_attr.shape[1] != self.num_edge_type:
                raise ValueError('edge_attr must be the one-hot [E, num_edge_type] tensor the collation emits')
            csr = ops.Csr.from_edge_index(edge_index, edge_attr, x.shape[0])
        mlp = self.conv.nn.seq
        fused = (isinstance(mlp, nn.Sequential) and len(mlp) == 3 and isinstance(mlp[1], nn.ReLU)
                 and mlp[0].bias is not None and mlp[2].bias is not None and self.node_trans.bias is not None)
        hidden = self.node_trans.out_features
        if isinstance(x, ops.EmbeddedIds):
            if fused and hidden % 4 == 0:
                # node_trans straight from the embedding ids: no [V, dim_in] rows, no product over them
                x0 = ops.embed_trans(x, self.node_trans.weight, self.node_trans.bias)
                out = ops.gine_encoder(x0, None, None, self.edge_table(), mlp[0].weight, mlp[0].bias,
                                       mlp[2].weight, mlp[2].bias, csr, self.t, not want_all)
                return (out[:, (self.t - 1) * hidden:], out) if want_all else (out, None)
            x = x.rows()
        if fused:
            # the reference's configuration (MLP(H, H, H, ReLU, 2), encoder.py:48): one fused node
            out = ops.gine_encoder(x.float(), self.node_trans.weight, self.node_trans.bias, self.edge_table(),
                                   mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, csr, self.t, not want_all)
            return (out[:, (self.t - 1) * hidden:], out) if want_all else (out, None)
        x = run_linear(self.node_trans, x.float())
        table = self.edge_table()
        all_x = []
        for _ in range(self.t):
            x = self.conv(x, csr, table)
            all_x.append(x)
        return x, (torch.cat(all_x, dim=-1) if want_all else None)

    def embed_graph_of_nodes(self, x, csr, segments):
        """embed_graph(embed_node(x)[1]) (src/modules/encoder.py:53-69) for batches in the compact
        layout: with the reference's conv MLP the sum over each graph's nodes is taken inside the
        encoder's autograd node, whose backward then works from one gradient row per graph."""
        mlp = self.conv.nn.seq
        fused = (isinstance(mlp, nn.Sequential) and len(mlp) == 3 and isinstance(mlp[1], nn.ReLU)
                 and mlp[0].bias is not None and mlp[2].bias is not None and self.node_trans.bias is not None)
        hidden = self.node_trans.out_features
        if (fused and hidden % 4 == 0 and isinstance(x, ops.EmbeddedIds) and isinstance(csr, ops.Csr)
                and isinstance(segments, ops.Segments) and segments.covers_all_rows):
            x0 = ops.embed_trans(x, self.node_trans.weight, self.node_trans.bias)
            pooled = ops.gine_encoder(x0, None, None, self.edge_table(), mlp[0].weight, mlp[0].bias,
                                      mlp[2].weight, mlp[2].bias, csr, self.t, False, segments)
            return run_linear(self.linear, pooled)
        _, all_x = self.embed_node(x, csr)
        return self.embed_graph(all_x, segments)

    def embed_graph(self, all_x, graph_ids, node_mask=None):
        if isinstance(graph_ids, ops.Segments):
            segs = graph_ids
        else:
            segs = _segments_from_graph_ids(graph_ids, node_mask)
        return run_linear(self.linear, ops.segment_sum(all_x, segs))

    def forward(self, batch, return_x=False):
        x, all_x = self.embed_node(batch.x, batch.edge_index, batch.edge_attr)
        res = self.embed_graph(all_x, batch.batch)
        if return_x:
            return res, x
        return res


def _segments_from_graph_ids(graph_ids, node_mask):
    """Row ranges of the graphs for non-decreasing graph_ids (what the reference passes:
    repeat_interleave(arange(B), N), src/pl_models/ps_vae_model.py:39); with a node mask the
    kept rows of a graph must be a prefix of its rows (padding sits at the end)."""
    num_graphs = int(graph_ids[-1]) + 1
    counts = torch.bincount(graph_ids, minlength=num_graphs)
    begin = torch.zeros(num_graphs, dtype=torch.int64, device=graph_ids.device)
    begin[1:] = torch.cumsum(counts, 0)[:-1]
    if node_mask is None:
        kept = counts
    else:
        kept = torch.bincount(graph_ids[~node_mask], minlength=num_graphs)
    if not bool((graph_ids[1:] >= graph_ids[:-1]).all()):
        raise ValueError('graph_ids must be non-decreasing')
    if node_mask is not None:
        local = torch.arange(graph_ids.numel(), device=graph_ids.device) - begin[graph_ids]
        if not bool(((local < kept[graph_ids]) == ~node_mask).all()):
            raise ValueError('masked rows must be the trailing rows of each graph')
    begin32 = begin.to(torch.int32)
    covers = node_mask is None
    return ops.Segments(begin32, (begin + kept).to(torch.int32), covers)
