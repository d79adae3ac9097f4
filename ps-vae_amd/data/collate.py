"""Batch collation on the GPU: host-side planning around the two collation kernels.

Replaces the DataLoader's __getitem__ + collate_fn of the reference
(src/data/bpe_dataset.py:56-155, 173-181).  The host only prefix-sums per-molecule counts
(a few numpy cumsums over the batch); all per-atom / per-cell work runs in
psvae_collate_count / psvae_collate_fill.  One device->host read per batch returns P, the
number of kept candidate pairs, which sizes golden_edge exactly as the reference does (in the
compact-only layout it is read late, off the critical path).
"""
import numpy as np
import torch

from .. import hip, ops

KEEP_PROB = 0.022   # src/data/bpe_dataset.py:129


class CompactBatch:
    """Real atoms only; node id = node_off[b] + m.  See include/psvae_hip.h (psvae_collate_fill).

    `pairs` / `golden` may be backed by buffers sized to an upper bound of P: the exact count
    is then read from the device (one int32, copied asynchronously during collation) the first
    time either is touched, which is after the encoder has been enqueued."""

    def __init__(self, t, node_off, pair_off, num_graphs, pending=None):
        self.cx, self.cxp, self.cxpos = t['cx'], t['cxp'], t['cxpos']
        self.node_off, self.pair_off = node_off, pair_off
        self.csr = ops.Csr(t['row_ptr'], t['col'], t['cls'])
        self.csr_in = ops.Csr(t['in_row_ptr'], t['in_col'], t['in_cls'])
        self.segments = ops.Segments.from_offsets(node_off)
        self._pair_bufs = (t['pair_src'], t['pair_dst'], t['pair_graph'], t['golden32'])
        self._pending = pending          # (event, pinned int32[1]) while P is still on its way
        self._pairs = self._golden = None
        self.num_graphs = num_graphs
        if pending is None:
            self._resolve(t['pair_src'].numel())

    def _resolve(self, P=None):
        if self._pairs is None:
            if P is None:
                event, host = self._pending
                event.synchronize()
                P = int(host[0])
                self._pending = None
                _PINNED_FREE.append(host)      # the staging buffer is free for the next batch
            src, dst, graph, gold = (b[:P] for b in self._pair_bufs)
            self._pairs = ops.PairList(src, dst, graph, self.pair_off, self.segments)
            self._golden = gold

    @property
    def pairs(self):
        self._resolve()
        return self._pairs

    @property
    def golden(self):
        self._resolve()
        return self._golden

    @property
    def num_nodes(self):
        return self.cx.numel()


_PINNED_FREE = []


def _pinned_int():
    """A pinned int32[1] host buffer that belongs to ONE batch until its pair count has been
    read (CompactBatch._resolve hands it back): however many batches are collated ahead of
    their use, none of them can read another batch's count.  A batch dropped unread simply
    keeps its 4 bytes."""
    return _PINNED_FREE.pop() if _PINNED_FREE else torch.empty(1, dtype=torch.int32).pin_memory()


class CollatedBatch(dict):
    """The reference's batch dict (keys of src/data/bpe_dataset.py:142-155, present when the
    reference layout was requested) plus `.compact` (CompactBatch or None), `.pieces`,
    `.props`, `.batch_size`."""
    compact = None


def plan(host, ids):
    """Shapes and prefix sums of a batch from the per-molecule counts (host numpy)."""
    ids = np.asarray(ids, dtype=np.int64)
    n_atoms = (host.atom_off[ids + 1] - host.atom_off[ids]).astype(np.int64)
    n_pieces = (host.piece_off[ids + 1] - host.piece_off[ids]).astype(np.int64)
    B = len(ids)
    offs = np.zeros((5, B + 1), dtype=np.int64)
    offs[0, 1:] = np.cumsum(n_atoms)
    offs[1, 1:] = np.cumsum(host.n_edge[ids])
    offs[2, 1:] = np.cumsum(host.n_inpiece[ids])
    offs[3, 1:] = np.cumsum(host.n_rng[ids])
    offs[4, :B] = ids
    if offs.max(initial=0) >= 2 ** 31:
        raise ValueError('batch too large for int32 offsets')
    return {'B': B, 'N': int(n_atoms.max(initial=0)), 'S': int(n_pieces.max(initial=0)),
            'V': int(offs[0, -1]), 'E': int(offs[1, -1]), 'Ein': int(offs[2, -1]), 'R': int(offs[3, -1]),
            'offs': offs.astype(np.int32)}


def collate_packed(dpacked, ids, uniforms=None, atom_pad=10, piece_pad=300, num_bond_type=4,
                   keep_prob=KEEP_PROB, reference_layout=True, compact_layout=True, generator=None):
    """Collate molecules `ids` of a DevicePackedMolecules.

    uniforms: float64 device tensor with at least R = sum(n_rng) entries, consumed in the
    order the reference calls random() (molecule, row, column over non-bonded candidate
    pairs).  None draws them with torch.rand on the device (optionally from `generator`).
    """
    dev = dpacked.device
    p = plan(dpacked.host, ids)
    B, N, S, V, E, Ein, R = (p[k] for k in ('B', 'N', 'S', 'V', 'E', 'Ein', 'R'))
    if N > 250:
        raise ValueError(f'a molecule of {N} atoms exceeds PSVAE_MAX_ATOMS (250)')
    offs = torch.from_numpy(p['offs']).to(dev, non_blocking=True)
    node_off, edge_off, inedge_off, rng_off, mol_ids = offs[0], offs[1], offs[2], offs[3], offs[4]
    if uniforms is None:
        uniforms = torch.rand(max(R, 1), dtype=torch.float64, device=dev, generator=generator)
    elif uniforms.numel() < R:
        raise ValueError(f'{R} uniforms needed, {uniforms.numel()} given')
    if uniforms.dtype != torch.float64 or not uniforms.is_cuda:
        raise TypeError('uniforms must be a float64 device tensor')
    pair_cnt = torch.empty(max(B, 1), dtype=torch.int32, device=dev)
    pair_off = torch.empty(B + 1, dtype=torch.int32, device=dev)
    hip.collate_count(B, N, mol_ids, dpacked, rng_off, uniforms, keep_prob, pair_cnt, pair_off)
    pending = None
    if reference_layout or not compact_layout:
        P = int(pair_off[B].item())      # golden_edge is sized exactly, as the reference's is
    else:
        # compact layout only: the pair outputs are sized to a bound known on the host (every
        # drawn uniform kept + every bonded cross-piece cell) and the exact count follows
        # asynchronously, so the host does not stop here
        P = R + (E - Ein)
        host_p = _pinned_int()
        host_p.copy_(pair_off[B:B + 1], non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        pending = (event, host_p)

    i32, i64, u8 = torch.int32, torch.int64, torch.uint8

    backing = {}

    class _Out(dict):
        # A zero-element tensor reports a NULL data pointer, which the C ABI reads as "skip
        # this output": every output is backed by at least one element and the backing
        # tensor is what crosses the ABI; the dict holds the exact-shape views.
        def __setitem__(self, key, shape_dtype):
            shape, dtype = shape_dtype
            n = int(np.prod(shape))
            buf = torch.empty(max(n, 1), dtype=dtype, device=dev)
            backing[key] = buf
            dict.__setitem__(self, key, buf[:n].view(*shape))

    def alloc(shape, dtype):
        return shape, dtype

    ref, cmp_ = _Out(), _Out()
    ref['pieces'] = alloc((B, S), i64)
    ref['props'] = alloc((B, 3), torch.float32)
    if reference_layout:
        for k in ('x', 'x_pieces', 'x_pos'):
            ref[k] = alloc((B, N), i64)
        ref['atom_mask'] = alloc((B, N), u8)
        ref['edge_index'] = alloc((2, E), i64)
        ref['edge_attr'] = alloc((E, num_bond_type), i64)
        ref['edge_select'] = alloc((B, N, N), u8)
        ref['golden_edge'] = alloc((P,), i64)
        ref['in_piece_edge_idx'] = alloc((Ein,), i64)
    if compact_layout:
        for k in ('cx', 'cxp', 'cxpos'):
            cmp_[k] = alloc((V,), i32)
        cmp_['row_ptr'] = alloc((V + 1,), i32)
        cmp_['col'] = alloc((E,), i32)
        cmp_['cls'] = alloc((E,), u8)
        cmp_['in_row_ptr'] = alloc((V + 1,), i32)
        cmp_['in_col'] = alloc((Ein,), i32)
        cmp_['in_cls'] = alloc((Ein,), u8)
        for k in ('pair_src', 'pair_dst', 'pair_graph', 'golden32'):
            cmp_[k] = alloc((P,), i32)
        if B == 0 or V == 0:
            cmp_['row_ptr'].zero_()
            cmp_['in_row_ptr'].zero_()
    hip.collate_fill(B, N, S, mol_ids, dpacked,
                     {'node_off': node_off, 'edge_off': edge_off, 'inedge_off': inedge_off,
                      'rng_off': rng_off, 'pair_off': pair_off},
                     E, uniforms, keep_prob, atom_pad, piece_pad, num_bond_type,
                     {k: backing[k] for k in ref}, {k: backing[k] for k in cmp_})
    out = CollatedBatch()
    if reference_layout:
        out.update(ref)
        out['atom_mask'] = ref['atom_mask'].view(torch.bool)
        out['edge_select'] = ref['edge_select'].view(torch.bool)
    out.pieces, out.props, out.batch_size = ref['pieces'], ref['props'], B
    out.num_uniforms = R
    if compact_layout:
        out.compact = CompactBatch(cmp_, node_off, pair_off, B, pending)
    return out
