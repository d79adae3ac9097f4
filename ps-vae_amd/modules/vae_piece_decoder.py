"""VAE piece decoder with the reference's constructor, methods and state_dict keys
(src/modules/vae_piece_decoder.py:16-126, and the link-prediction slice of inference,
:203-241).

What runs where: embeddings, the second GINE pass, the pair gather, the link MLP, the link
loss, the reparameterised latent and the teacher-forced GRU piece language model (:81-90)
are HIP kernels (ps_vae_amd.ops); `self.rnn` stays an nn.GRU so that the parameter names
match, but only its weights are used.  The RDKit-based molecule assembly of `inference`
(:160-201, 243-274) is outside the hot path."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip, ops
from ..utils.chem_vocab import MAX_VALENCE
from .common_nn import MLP, run_linear, run_sequential
from .encoder import Encoder


_SIDE_STREAMS = {}   # one extra stream per device, shared by every decoder instance
_ZERO_ROWS = {}


def _zero_rows(rows, width, device):
    """A cached [rows, width] block of zeros (padding rows of a weight; never written)."""
    key = (rows, width, device)
    if key not in _ZERO_ROWS:
        _ZERO_ROWS[key] = torch.zeros(rows, width, dtype=torch.float32, device=device)
    return _ZERO_ROWS[key]


class VAEPieceDecoder(nn.Module):
    def __init__(self, atom_embedding_dim, piece_embedding_dim, max_pos, pos_embedding_dim, piece_hidden_dim,
                 node_hidden_dim, num_edge_type, cond_dim, latent_dim, tokenizer, t=4):
        super(VAEPieceDecoder, self).__init__()
        self.tokenizer = tokenizer
        self.atom_embedding = nn.Embedding(tokenizer.num_atom_type(), atom_embedding_dim)
        self.piece_embedding = nn.Embedding(tokenizer.num_piece_type(), piece_embedding_dim)
        self.pos_embedding = nn.Embedding(max_pos, pos_embedding_dim)
        self.latent_to_rnn_hidden = nn.Linear(latent_dim, piece_hidden_dim)
        self.rnn = nn.GRU(piece_embedding_dim, piece_hidden_dim, batch_first=True)
        self.to_vocab = nn.Linear(piece_hidden_dim, tokenizer.num_piece_type())
        node_dim = atom_embedding_dim + piece_embedding_dim + pos_embedding_dim
        self.graph_embedding = Encoder(node_dim, num_edge_type, node_hidden_dim, dim_out=1, t=t)
        mlp_in = node_hidden_dim * 2 + latent_dim
        self.edge_predictor = nn.Sequential(
            MLP(dim_in=mlp_in, dim_hidden=mlp_in // 2, dim_out=mlp_in, act_func=nn.ReLU, num_layers=3),
            nn.Linear(mlp_in, num_edge_type))
        self.latent_dim = latent_dim
        self.W_mean = nn.Linear(cond_dim, latent_dim)
        self.W_log_var = nn.Linear(cond_dim, latent_dim)
        self.piece_loss = nn.CrossEntropyLoss(ignore_index=tokenizer.pad_idx())
        self.edge_loss = nn.CrossEntropyLoss()   # kept for API parity; the fused kernel computes it

    def rsample(self, conds, eps=None):
        """:56-63.  `eps` injects the normal draw (the reference calls torch.randn_like)."""
        z_mean = run_linear(self.W_mean, conds)
        z_log_var_pre = run_linear(self.W_log_var, conds)
        if eps is None:
            eps = torch.randn_like(z_mean)
        return ops.rsample_kl(z_mean, z_log_var_pre, eps)

    def embedded_ids(self, atom_ids, piece_ids, pos_ids):
        """The atoms as ids + tables (ops.EmbeddedIds): what embed_atom would return, for
        encoders that compute their first layer from the ids."""
        flat = [t.reshape(-1).to(torch.int32) for t in (atom_ids, piece_ids, pos_ids)]
        ids = ops.EmbeddedIds(flat[0], flat[1], flat[2], self.atom_embedding.weight, self.piece_embedding.weight,
                              self.pos_embedding.weight)
        if torch.is_grad_enabled() and flat[0].numel():
            ids.sorted_keys()   # the backward of both encoders reads it, possibly on two streams
        return ids

    def embed_atom(self, atom_ids, piece_ids, pos_ids):
        """:65-69 — ids of any shape -> [..., atom+piece+pos dims]."""
        shape = atom_ids.shape
        flat = [t.reshape(-1).to(torch.int32) for t in (atom_ids, piece_ids, pos_ids)]
        out = ops.embed_atom(flat[0], flat[1], flat[2], self.atom_embedding.weight,
                             self.piece_embedding.weight, self.pos_embedding.weight)
        return out.view(*shape, out.shape[-1])

    def piece_logits(self, z_vecs, pieces, piece_table=None, time_major=False, pad_vocab=False):
        """:81-90 — teacher-forced GRU over the piece sequence, initial state from z.
        piece_table: self.piece_embedding.weight as the caller wants it seen by autograd (the
        two-stream schedule hands it over from the main stream).  Returns [B, S-1, vocab]; with
        time_major=True the same logits as the GRU path produces them, [S-1, B, vocab] (no
        transposition: a loss that is a mean over tokens does not care about their order).
        pad_vocab (time-major GRU path only): the vocabulary layer gets zero rows up to a multiple of
        four classes, so [S-1, B, vocab rounded up]: the three products around the logits then take
        the 16-byte loaders (302 classes do not); the loss kernel is told the real class count."""
        rnn = self.rnn
        init_hidden = run_linear(self.latent_to_rnn_hidden, z_vecs)
        if rnn.num_layers == 1 and not rnn.bidirectional and rnn.bias:
            # time-major ids -> [T,B,E] embeddings -> HIP GRU
            ids = pieces[:, :-1].t().to(torch.int32).contiguous()      # (one kernel: the cast writes the transposed rows)
            table = self.piece_embedding.weight if piece_table is None else piece_table
            emb = ops.embedding(ids, table) if table.shape[1] % 4 == 0 else self.piece_embedding(ids.long())
            hs = ops.gru(emb, init_hidden, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0)
            pad = (-self.to_vocab.out_features) % 4 if (pad_vocab and time_major and self.to_vocab.bias is not None) else 0
            if pad:
                zeros = _zero_rows(pad, self.to_vocab.in_features, hs.device)
                out = ops.linear(hs, torch.cat([self.to_vocab.weight, zeros], dim=0),
                                 torch.cat([self.to_vocab.bias, zeros[:, 0]], dim=0))
            else:
                out = run_linear(self.to_vocab, hs)
            return out if time_major else out.transpose(0, 1)
        output, _ = rnn(self.piece_embedding(pieces)[:, :-1], init_hidden.unsqueeze(0))
        out = run_linear(self.to_vocab, output)
        return out.transpose(0, 1) if time_major else out

    def piece_loss_fused(self, z_vecs, pieces, piece_table=None):
        """:81-90 + :113-114 (+ the counts behind :120-122): the piece language model's logits and
        their cross-entropy against pieces[:, 1:] with <pad> ignored, in one kernel on the
        time-major logits.  Returns (loss, counts int32 [non-pad tokens, correct argmax])."""
        logits = self.piece_logits(z_vecs, pieces, piece_table, time_major=True, pad_vocab=True)
        gold = pieces[:, 1:].t().to(torch.int32).contiguous()
        return ops.token_ce(logits, gold, self.piece_loss.ignore_index, n_classes=self.to_vocab.out_features)

    def link_logits(self, node_embedding, z_vecs, pairs, golden=None):
        """:99-110 (+ :115 when `golden` is given).  node_embedding [V,H] in the node ids the
        pair list uses.  Returns (logits, loss or None, n_correct or None)."""
        mlp, last = self.edge_predictor[0], self.edge_predictor[1]
        seq = mlp.seq
        H, L = node_embedding.shape[-1], z_vecs.shape[-1]
        split_first = (isinstance(seq, nn.Sequential) and len(seq) >= 4 and isinstance(seq[0], nn.Linear)
                       and isinstance(seq[1], nn.ReLU) and isinstance(seq[-1], nn.Linear) and seq[0].bias is not None
                       and seq[0].in_features == 2 * H + L and seq[0].out_features % 4 == 0 and H % 4 == 0)
        if split_first:
            # The first layer's weight is cut at the [h_src | h_dst | z] blocks and applied once per
            # atom / per molecule (one product for both atom blocks, a small one for z); per pair only
            # a gather-add-ReLU remains, and neither the [P, 2H+L] feature rows nor their gradient are
            # ever formed.  In training P ~ V, so the arithmetic is the same; over ALL candidate pairs
            # of generated molecules (pairs >> atoms) it is several times less.
            first = seq[0]
            w_a, w_b, w_z = ops.split_columns(first.weight, (H, H, L))
            w_nodes = torch.cat([w_a, w_b], dim=0)                     # [2F, H]: rows of Wa, then of Wb
            huv = ops.linear(node_embedding, w_nodes)                  # [V, 2F]
            hz = ops.linear(z_vecs, w_z, first.bias)                   # [B, F]
            hid = ops.pair_combine(huv, hz, pairs, relu=True)
            weight, bias = ops.affine_merge(last.weight, last.bias, seq[-1].weight, seq[-1].bias)
            inner = seq[2:-1]
            if (golden is None and not torch.is_grad_enabled() and len(inner) == 2 and isinstance(inner[0], nn.Linear)
                    and isinstance(inner[1], nn.ReLU)):
                # generation: hidden Linear + ReLU + class layer as one product whose [P, F] result is never
                # written (the class layer runs in the product's epilogue)
                logits = ops.linear_relu_classes(hid, inner[0].weight, inner[0].bias, weight, bias)
                if logits is not None:
                    return logits, None, None
            hid = run_sequential(inner, hid)
            if golden is None:
                return ops.linear(hid, weight, bias), None, None
            loss, logits, n_ok = ops.link_ce(hid, weight, bias, golden, getattr(pairs, 'live', None))
            return logits, loss, n_ok
        feats = ops.pair_gather(node_embedding, z_vecs, pairs)
        if isinstance(seq, nn.Sequential) and len(seq) > 1 and isinstance(seq[-1], nn.Linear):
            # No activation sits between the MLP's output layer and the class layer, so the
            # two are one affine map: W = W_last W_out, b = W_last b_out + b_last.  The wide
            # intermediate [P, mlp_in] is never formed; the gradient of (W, b) goes back to both
            # layers through one small kernel.
            inner = seq[-1]
            hid = run_sequential(seq[:-1], feats)
            weight, bias = ops.affine_merge(last.weight, last.bias, inner.weight, inner.bias)
        else:
            hid = mlp(feats)
            weight, bias = last.weight, last.bias
        if golden is None:
            return ops.linear(hid, weight, bias), None, None
        loss, logits, n_ok = ops.link_ce(hid, weight, bias, golden, getattr(pairs, 'live', None))
        return logits, loss, n_ok

    def embed_nodes_async(self, node_x, csr):
        """Message passing over the in-piece bonds (:96) on a side stream; returns a handle
        for forward(node_embedding=...), which joins the streams before the first use."""
        ids = node_x if isinstance(node_x, ops.EmbeddedIds) else None
        main = torch.cuda.current_stream(node_x.device)
        side = _SIDE_STREAMS.get(node_x.device)
        if side is None:
            side = _SIDE_STREAMS[node_x.device] = torch.cuda.Stream(node_x.device)
        if ids is not None:
            node_x = ids.for_side_stream()   # the embedding tables are shared with the main-stream encoder
        side.wait_stream(main)
        with torch.cuda.stream(side):
            node_embedding, _ = self.graph_embedding.embed_node(node_x, csr, None, want_all=False)
        for t in (ids.tensors() if ids is not None else (node_x,)):
            t.record_stream(side)
        return node_embedding, side

    def forward(self, x, x_pieces, x_pos, edge_index, edge_attr, pieces, conds, edge_select, golden_edge,
                return_accu=False, eps=None, node_embedding=None):
        """:71-126.  Accepts the reference's tensors (x [B,N,D], edge_index/edge_attr of the
        in-piece edges, edge_select [B,N,N] bool, golden_edge [P] int64) or the compact
        containers (x [V,D], edge_index = ops.Csr, edge_select = ops.PairList)."""
        z_vecs, kl_loss = self.rsample(conds, eps)
        side = node_embedding[1] if node_embedding is not None else None

        def piece_branch(piece_table=None):
            return self.piece_loss_fused(z_vecs, pieces, piece_table)

        if side is not None:
            # The piece language model (many small, latency-bound launches) shares the chip with
            # the link predictor's large products: it continues on the side stream, behind the
            # message passing that is already queued there.
            main = torch.cuda.current_stream(z_vecs.device)
            node_embedding = node_embedding[0]
            main.wait_stream(side)                  # node_embedding is complete
            node_embedding.record_stream(main)
            # the piece table also feeds the atom features of the main-stream encoder
            piece_table = ops.stream_handoff(self.piece_embedding.weight)
            side.wait_stream(main)                  # z_vecs is complete
            with torch.cuda.stream(side):
                piece_loss, piece_counts = piece_branch(piece_table)
            for t in (z_vecs, pieces):
                t.record_stream(side)
        else:
            node_x = x if isinstance(x, ops.EmbeddedIds) else x.reshape(-1, x.shape[-1])
            node_embedding, _ = self.graph_embedding.embed_node(node_x, edge_index, edge_attr, want_all=False)
        pairs = edge_select if isinstance(edge_select, ops.PairList) else ops.PairList.from_edge_select(edge_select)
        golden = golden_edge if golden_edge.dtype == torch.int32 else golden_edge.to(torch.int32)
        pred_edge, edge_loss, n_ok = self.link_logits(node_embedding, z_vecs, pairs, golden)

        if side is not None:
            main.wait_stream(side)
            piece_counts.record_stream(main)
            piece_loss.record_stream(main)
        else:
            piece_loss, piece_counts = piece_branch()
        rec_loss = piece_loss + edge_loss
        loss_tuple = (rec_loss, piece_loss, edge_loss, kl_loss)
        if return_accu:
            n_tok, n_hit = piece_counts.tolist()     # non-pad tokens, those whose argmax is the gold piece
            piece_accu = n_hit / n_tok
            edge_accu = n_ok.item() / len(golden)
            return z_vecs, (loss_tuple, (piece_accu, edge_accu))
        return z_vecs, loss_tuple

    @torch.no_grad()
    def infer_links(self, z, x, x_pieces, x_pos, edge_index, edge_attr, groups, add_edge_th):
        """Link prediction over ALL candidate pairs of generated fragments (:203-241):
        candidates = upper triangle minus pairs inside one fragment, softmax, (max prob,
        argmax), keep conf >= add_edge_th and type != None, order by confidence descending
        (stable).  `groups`: list of atom-id lists, one per fragment.
        Returns (pairs [Ec,2] int64, confidence [Ec], edge_type [Ec] int64, sorted_idx list)."""
        dev = self.W_mean.weight.device
        n = x.numel()
        node_x = self.embed_atom(x.to(dev), x_pieces.to(dev), x_pos.to(dev))
        h, _ = self.graph_embedding.embed_node(node_x, edge_index.to(dev), edge_attr.to(dev), want_all=False)
        frag = torch.full((n,), -1, dtype=torch.int64)
        for gi, g in enumerate(groups):
            frag[torch.as_tensor(g, dtype=torch.int64)] = gi
        frag = frag.to(dev)
        sel = torch.triu(torch.ones(n, n, dtype=torch.bool, device=dev))
        sel &= ~((frag[:, None] == frag[None, :]) & (frag[:, None] >= 0))
        pairs = ops.PairList.from_edge_select(sel.unsqueeze(0))
        idx = torch.stack([pairs.src, pairs.dst], dim=1).to(torch.int64)
        if len(pairs) == 0:
            return idx, torch.zeros(0, device=dev), torch.zeros(0, dtype=torch.int64, device=dev), []
        logits, _, _ = self.link_logits(h, z.to(dev).view(1, -1), pairs)
        P = logits.shape[0]
        conf = torch.empty(P, dtype=torch.float32, device=dev)
        etype = torch.empty(P, dtype=torch.int32, device=dev)
        keep = torch.empty(P, dtype=torch.uint8, device=dev)
        hip.link_infer(logits.contiguous(), add_edge_th, self.tokenizer.chem_vocab.bond_to_idx(None), conf, etype, keep)
        kept = keep.bool().nonzero().flatten()
        order = kept[torch.argsort(conf[kept], descending=True, stable=True)]
        return idx, conf, etype.to(torch.int64), order.tolist()

    @torch.no_grad()
    def sample_pieces_batch(self, z, max_atom_num, temperature, uniforms=None, piece_atom_counts=None,
                            generator=None):
        """The piece sampling loop of `inference` (:134-158) for a batch of latents: every step
        is one GRU cell over all still-active molecules (two GEMMs + the fused gate kernel),
        the vocabulary projection, the pad / first-step end masks, softmax(logits / T) and one
        draw per molecule.  The draw is the inverse-CDF form of torch.multinomial(probs, 1):
        piece = first index whose cumulative probability exceeds uniforms[b, t] x total (uniforms
        [B, max_pos] in [0,1), drawn here when None).  A molecule stops after the end token,
        when its atom count exceeds max_atom_num or at max_pos pieces, and its last sampled
        piece is dropped (:154-158).  Returns (piece_ids [B, max_pos-ish] padded with pad_idx,
        lengths [B])."""
        dev = self.W_mean.weight.device
        tok = self.tokenizer
        z = z.to(dev).float().reshape(-1, self.latent_dim)
        B = z.shape[0]
        max_pos = self.pos_embedding.num_embeddings
        pad, end = tok.pad_idx(), tok.end_idx()
        if piece_atom_counts is None:
            piece_atom_counts = tok.piece_atom_counts()
        cnt = torch.as_tensor(piece_atom_counts, dtype=torch.int64, device=dev)
        if uniforms is None:
            uniforms = torch.rand(B, max_pos, device=dev, generator=generator)
        uniforms = uniforms.to(dev).float()
        rnn = self.rnn
        H = rnn.hidden_size
        w_ih, w_hh = rnn.weight_ih_l0.contiguous(), rnn.weight_hh_l0.contiguous()
        E = w_ih.shape[1]
        h = run_linear(self.latent_to_rnn_hidden, z).contiguous()
        h_next = torch.empty_like(h)
        gi = torch.empty(B, 3 * H, device=dev)
        gh = torch.empty(B, 3 * H, device=dev)
        gates = torch.empty(B, 3 * H, device=dev)
        cur = torch.full((B,), end, dtype=torch.int64, device=dev)
        out = torch.full((B, max_pos), pad, dtype=torch.int64, device=dev)
        length = torch.zeros(B, dtype=torch.int32, device=dev)
        atoms = torch.zeros(B, dtype=torch.int32, device=dev)
        active = torch.ones(B, dtype=torch.uint8, device=dev)
        cnt = cnt.to(torch.int32)
        for t in range(max_pos):
            emb = self.piece_embedding(cur).contiguous()
            hip.gemm(B, 3 * H, E, emb, E, 1, w_ih, 1, E, gi, 3 * H, bias=rnn.bias_ih_l0)
            hip.gemm(B, 3 * H, H, h, H, 1, w_hh, 1, H, gh, 3 * H, bias=rnn.bias_hh_l0)
            hip.gru_cell_fwd(gi, gh, h, h_next, gates)
            logits = run_linear(self.to_vocab, h_next)
            # pad / first-step end masks, softmax(logits / T), the draw and the state update: one kernel
            hip.piece_draw(logits, temperature, uniforms[:, t], pad, end, t == 0, cnt, max_atom_num, max_pos, active, atoms,
                           length, out[:, t], cur)
            h, h_next = h_next, h
            if t % 8 == 7 and not bool(active.any()):
                break
        length = length.to(torch.int64)
        lengths = (length - 1).clamp(min=0)
        keep = torch.arange(max_pos, device=dev)[None, :] < lengths[:, None]
        return torch.where(keep, out, torch.full_like(out, pad)), lengths

    @torch.no_grad()
    def infer_links_batch(self, z, x, x_pieces, x_pos, node_off, csr_in, frag, add_edge_th):
        """infer_links for many generated molecules at once (the link-prediction slice of
        :203-241 over a batch; the reference loops over molecules in Python,
        src/pl_models/ps_vae_model.py:139-141).

        z [B,L]; x / x_pieces / x_pos / frag: flat [V] per-atom ids of all molecules
        (molecule b owns atoms node_off[b] .. node_off[b+1]); csr_in: ops.Csr of the in-piece
        bonds over the flat node ids; frag: fragment id of every atom (any integers, distinct
        per fragment inside a molecule).  Candidates of a molecule are its atom pairs (m < n)
        in different fragments, in row-major order.

        Returns a dict: graph / m / n [Ec] (molecule and molecule-local atom ids of every
        candidate), conf, etype [Ec], ptr [B+1] (candidates of molecule b are ptr[b]..ptr[b+1]),
        order [Ek] (indices of the kept candidates: conf >= add_edge_th and type != None,
        grouped by molecule, confidence descending, stable) and order_ptr [B+1]."""
        dev = self.W_mean.weight.device
        no32 = node_off.to(dev).to(torch.int32).contiguous()
        B = no32.numel() - 1
        # candidate enumeration in two kernels (count per molecule, scan, fill): row-major (m, k), m < k, atoms
        # in different fragments
        max_atoms = int((no32[1:] - no32[:-1]).max().item()) if B > 0 else 0
        src, dst, graph32, m32, k32, ptr = hip.pair_enumerate(no32, frag.to(dev).to(torch.int32).contiguous(), max_atoms)
        graph, m, k = graph32.to(torch.int64), m32.to(torch.int64), k32.to(torch.int64)
        pairs = ops.PairList(src, dst, graph32, ptr, ops.Segments.from_offsets(no32))
        pairs.max_nodes = max_atoms
        out = {'graph': graph, 'm': m, 'n': k, 'ptr': ptr}
        P = src.numel()
        if P == 0:
            out.update(conf=torch.zeros(0, device=dev), etype=torch.zeros(0, dtype=torch.int64, device=dev),
                       order=torch.zeros(0, dtype=torch.int64, device=dev),
                       order_ptr=torch.zeros(B + 1, dtype=torch.int64, device=dev))
            return out
        node_x = self.embed_atom(x.to(dev), x_pieces.to(dev), x_pos.to(dev))
        h, _ = self.graph_embedding.embed_node(node_x, csr_in, None, want_all=False)
        logits, _, _ = self.link_logits(h, z.to(dev).view(B, -1), pairs)
        conf = torch.empty(P, dtype=torch.float32, device=dev)
        etype = torch.empty(P, dtype=torch.int32, device=dev)
        keep = torch.empty(P, dtype=torch.uint8, device=dev)
        hip.link_infer(logits.contiguous(), add_edge_th, self.tokenizer.chem_vocab.bond_to_idx(None), conf, etype, keep)
        kept = keep.bool().nonzero().flatten()
        by_conf = kept[torch.argsort(conf[kept], descending=True, stable=True)]
        order = by_conf[torch.argsort(graph[by_conf], stable=True)]
        order_ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        order_ptr[1:] = torch.cumsum(torch.bincount(graph[order], minlength=B), 0)
        out.update(conf=conf, etype=etype.to(torch.int64), order=order, order_ptr=order_ptr)
        return out

    @torch.no_grad()
    def assemble_links_batch(self, links, x, node_off, csr_in, block0, n_blocks):
        """The greedy bond loop of `inference` (:245-268) for many generated molecules at once:
        the kept candidates of `links` (the dict infer_links_batch returns) are visited per
        molecule in confidence order; a bond is added when valence_check and cycle_check
        (src/utils/chem_utils.py:447-483) allow it, fragments joined by a bond merge into one
        block and the atoms outside the largest block are marked for removal.

        x [V]: chemical-vocabulary id of every atom; csr_in: ops.Csr of the bonds inside the
        fragments over the flat atom ids (cls = bond index 1..3); block0 [V]: fragment index
        of every atom inside its molecule (0 .. n_blocks[b]-1, the order of `groups` at :189);
        n_blocks [B].

        Returns a dict: accepted [Ek] bool (aligned with links['order']), block [V] (final
        block id per atom), keep_atom [V] bool, status [B] (0 ok, 1 an atom exceeded 8
        neighbours, 2 bad ids)."""
        dev = self.W_mean.weight.device
        i32 = torch.int32
        vocab = self.tokenizer.chem_vocab
        node_off = node_off.to(dev).to(i32).contiguous()
        B = node_off.numel() - 1
        order = links['order']
        ord_m = links['m'][order].to(i32).contiguous()
        ord_n = links['n'][order].to(i32).contiguous()
        ord_type = links['etype'][order].to(i32).contiguous()
        order_ptr = links['order_ptr'].to(i32).contiguous()
        atom_type = x.to(dev).to(i32).contiguous()
        V = atom_type.numel()
        n_atom = len(MAX_VALENCE)
        max_valence = torch.tensor([MAX_VALENCE[vocab.idx_to_atom(i)] for i in range(n_atom)], dtype=i32, device=dev)
        accepted = torch.empty(ord_m.numel(), dtype=torch.uint8, device=dev)
        block = torch.empty(V, dtype=i32, device=dev)
        keep_atom = torch.empty(V, dtype=torch.uint8, device=dev)
        status = torch.zeros(B, dtype=i32, device=dev)
        hip.assemble_links(node_off, atom_type, block0.to(dev).to(i32).contiguous(),
                           n_blocks.to(dev).to(i32).contiguous(), csr_in.row_ptr, csr_in.col, csr_in.cls,
                           order_ptr, ord_m, ord_n, ord_type, max_valence, vocab.atom_to_idx('S'),
                           accepted, block, keep_atom, status)
        return {'accepted': accepted.bool(), 'block': block, 'keep_atom': keep_atom.bool(), 'status': status}

    @torch.no_grad()
    def expand_pieces(self, piece_ids, lengths, fragments=None):
        """Sampled pieces -> the flat fragment graph of every molecule (:160-200 for a batch):
        atom types, piece id and position (+1, as :183) of every atom, the in-piece bonds as a
        CSR and the fragment index of every atom, all gathered from the vocabulary's fragment
        table on the device (no per-molecule Python).  piece_ids [B,T] padded, lengths [B].
        Returns a dict: x, x_pieces, x_pos, block0 [V]; node_off [B+1]; n_blocks [B]; csr_in."""
        dev = self.W_mean.weight.device
        table = (fragments or self.tokenizer.fragment_table()).on(dev)
        i64 = torch.int64
        piece_ids, lengths = piece_ids.to(dev), lengths.to(dev).to(i64)
        B, T = piece_ids.shape
        valid = torch.arange(T, device=dev)[None, :] < lengths[:, None]
        mol_of_piece, pos_of_piece = valid.nonzero(as_tuple=True)            # row-major: molecule, then position
        pid = piece_ids[valid].to(i64)
        a0, e0 = table['atom_off'].to(i64), table['edge_off'].to(i64)
        na, ne = a0[pid + 1] - a0[pid], e0[pid + 1] - e0[pid]
        V, E = int(na.sum().item()), int(ne.sum().item())
        piece_atom_base = torch.cumsum(na, 0) - na                            # first flat atom of each piece instance
        piece_edge_base = torch.cumsum(ne, 0) - ne
        inst = torch.repeat_interleave(torch.arange(pid.numel(), device=dev), na, output_size=V)
        local = torch.arange(V, device=dev) - piece_atom_base[inst]
        x = table['atom_type'].to(i64)[a0[pid][inst] + local]
        x_pieces, block0 = pid[inst], pos_of_piece[inst]
        atoms_per_mol = torch.zeros(B, dtype=i64, device=dev).index_add_(0, mol_of_piece, na)
        node_off = torch.zeros(B + 1, dtype=i64, device=dev)
        node_off[1:] = torch.cumsum(atoms_per_mol, 0)
        einst = torch.repeat_interleave(torch.arange(pid.numel(), device=dev), ne, output_size=E)
        elocal = torch.arange(E, device=dev) - piece_edge_base[einst]
        eidx = e0[pid][einst] + elocal
        src = piece_atom_base[einst] + table['edge_src'].to(i64)[eidx]
        dst = piece_atom_base[einst] + table['edge_dst'].to(i64)[eidx]
        row_ptr = torch.zeros(V + 1, dtype=i64, device=dev)
        row_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=V), 0)       # edges are already sorted by source
        csr_in = ops.Csr(row_ptr.to(torch.int32), dst.to(torch.int32).contiguous(),
                         table['edge_cls'][eidx].contiguous())
        return {'x': x, 'x_pieces': x_pieces, 'x_pos': block0 + 1, 'block0': block0, 'node_off': node_off,
                'n_blocks': lengths, 'csr_in': csr_in}

    @torch.no_grad()
    def inference_batch(self, z, max_atom_num, add_edge_th, temperature, uniforms=None, fragments=None):
        """`inference` (:128-270) for a batch of latents without its RDKit calls: sample the
        pieces, expand them into fragment graphs, predict the links between all cross-fragment
        atom pairs, add bonds greedily under the valence / ring-size rules and keep the largest
        connected block.  Returns (molecules, aux): molecules[b] = {'atoms': [chemical-vocabulary
        index], 'bonds': [(begin, end, bond index)]} with the dropped atoms removed and the rest
        renumbered (what the reference hands to Chem.SanitizeMol / Kekulize at :272-273, which
        stay with RDKit); aux holds the device-side intermediate results."""
        piece_ids, lengths = self.sample_pieces_batch(z, max_atom_num, temperature, uniforms=uniforms)
        g = self.expand_pieces(piece_ids, lengths, fragments)
        dev = g['x'].device
        B = lengths.numel()
        links = self.infer_links_batch(z, g['x'], g['x_pieces'], g['x_pos'], g['node_off'], g['csr_in'], g['block0'],
                                       add_edge_th)
        asm = self.assemble_links_batch(links, g['x'], g['node_off'], g['csr_in'], g['block0'], g['n_blocks'])
        # packaging on the host: kept atoms renumbered, in-piece bonds (one direction) + accepted links
        keep = asm['keep_atom']
        node_off = g['node_off']
        new_id = torch.cumsum(keep.to(torch.int64), 0) - 1
        mol_of_atom = torch.repeat_interleave(torch.arange(B, device=dev), node_off[1:] - node_off[:-1],
                                              output_size=keep.numel())
        kept_before = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        kept_before[1:] = torch.cumsum(torch.zeros(B, dtype=torch.int64, device=dev).index_add_(
            0, mol_of_atom, keep.to(torch.int64)), 0)
        csr = g['csr_in']
        V = keep.numel()
        src = torch.repeat_interleave(torch.arange(V, device=dev),
                                      (csr.row_ptr[1:] - csr.row_ptr[:-1]).to(torch.int64), output_size=csr.col.numel())
        dst = csr.col.to(torch.int64)
        one_dir = src < dst
        b_src, b_dst, b_cls = src[one_dir], dst[one_dir], csr.cls[one_dir].to(torch.int64)
        order = links['order']
        acc = asm['accepted']
        l_mol = links['graph'][order][acc]
        l_src = node_off[l_mol] + links['m'][order][acc]
        l_dst = node_off[l_mol] + links['n'][order][acc]
        l_cls = links['etype'][order][acc]
        all_src, all_dst = torch.cat([b_src, l_src]), torch.cat([b_dst, l_dst])
        all_cls = torch.cat([b_cls, l_cls])
        ok = keep[all_src] & keep[all_dst]
        all_src, all_dst, all_cls = all_src[ok], all_dst[ok], all_cls[ok]
        bond_mol = mol_of_atom[all_src]
        perm = torch.argsort(bond_mol, stable=True)
        bond_mol = bond_mol[perm]
        base = kept_before[bond_mol]
        bonds = torch.stack([new_id[all_src[perm]] - base, new_id[all_dst[perm]] - base, all_cls[perm]], 1).cpu().numpy()
        bond_cnt = torch.bincount(bond_mol, minlength=B).cpu().numpy()
        atoms = g['x'][keep].cpu().numpy()
        atom_cnt = (kept_before[1:] - kept_before[:-1]).cpu().numpy()
        mols, a, e = [], 0, 0
        for b in range(B):
            mols.append({'atoms': atoms[a:a + atom_cnt[b]].tolist(),
                         'bonds': [tuple(int(v) for v in row) for row in bonds[e:e + bond_cnt[b]]]})
            a += atom_cnt[b]
            e += bond_cnt[b]
        aux = dict(piece_ids=piece_ids, lengths=lengths, graph=g, links=links, assembly=asm)
        return mols, aux

    def inference(self, z, max_atom_num, add_edge_th, temperature):
        """:128-274 for one latent; returns the generated molecule as an atoms / bonds dict
        (the reference returns an rdkit Mol after SanitizeMol / Kekulize, which need RDKit)."""
        mols, _ = self.inference_batch(z.reshape(1, -1), max_atom_num, add_edge_th, temperature)
        return mols[0]
