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


# ---------------------------------------------------------------------------- fixed-size batches
class PinnedRing:
    """A few pinned host buffers used in turn for asynchronous host -> device copies: the host may run
    several steps ahead of the GPU, so a buffer is rewritten only after the copy that last read it
    has completed (its event is waited for when the ring comes round)."""

    def __init__(self, shape, dtype, slots=8):
        self.bufs = [torch.zeros(*shape, dtype=dtype).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.i = -1

    def next(self):
        self.i = (self.i + 1) % len(self.bufs)
        if self.events[self.i] is not None:
            self.events[self.i].synchronize()
        return self.bufs[self.i]

    def sent(self):
        """Call right after enqueueing the copy out of the buffer next() returned."""
        event = torch.cuda.Event()
        event.record()
        self.events[self.i] = event


class StaticShapes:
    """Sizes of a padded batch that do not change from batch to batch (what a step recorded once in
    a HIP graph needs): B molecules, V atom rows, E / Ein directed (in-piece) bonds, P candidate
    pairs, R uniforms, S pieces per molecule, N atoms of the largest molecule.  Rows beyond a
    batch's own counts are padding that takes no part in any result (see collate_static)."""

    def __init__(self, B, V, E, Ein, P, R, S, N):
        self.B, self.V, self.E, self.Ein, self.P, self.R, self.S, self.N = (int(v) for v in (B, V, E, Ein, P, R, S, N))

    def __repr__(self):
        return 'StaticShapes(' + ', '.join(f'{k}={getattr(self, k)}' for k in 'BVEPRSN') + f', Ein={self.Ein})'

    @classmethod
    def for_dataset(cls, host, B, sigmas=4.0, keep_prob=KEEP_PROB):
        """Bounds that a batch of B molecules drawn from `host` (PackedMolecules) exceeds with the
        probability of a `sigmas`-sigma event per quantity (mean x B + sigmas x std x sqrt(B), rounded
        up to a multiple of 32, never beyond the B largest molecules' total); a batch that does exceed
        one is collated and stepped the ordinary way.  S and N are the dataset's maxima."""
        n_atoms = np.diff(host.atom_off).astype(np.float64)

        def bound(per_mol):
            per_mol = np.asarray(per_mol, dtype=np.float64)
            stat = per_mol.mean() * B + sigmas * per_mol.std() * np.sqrt(B)
            worst = np.sort(per_mol)[::-1][:B].sum()
            return int(min(np.ceil(stat / 32.0) * 32, max(worst, 1)))

        cross_bonds = (host.n_edge - host.n_inpiece).astype(np.float64)       # bonded cross-piece cells (always kept)
        pairs = cross_bonds + keep_prob * host.n_rng                           # expected kept pairs per molecule
        pair_var = keep_prob * (1 - keep_prob) * host.n_rng                    # plus the binomial spread of the draws
        P = pairs.mean() * B + sigmas * np.sqrt((pairs.var() + pair_var.mean()) * B)
        return cls(B=B, V=bound(n_atoms), E=bound(host.n_edge), Ein=bound(host.n_inpiece),
                   P=int(np.ceil(P / 32.0) * 32), R=bound(host.n_rng), S=int(np.diff(host.piece_off).max()),
                   N=int(n_atoms.max()))


class StaticBatch:
    """The device buffers of one fixed-size batch, allocated once and refilled by collate_static:
    every kernel of a recorded step reads the same addresses on every replay."""

    def __init__(self, shapes, device, num_bond_type=4):
        s, dev = shapes, torch.device(device)
        i32, u8 = torch.int32, torch.uint8
        self.shapes, self.device = s, dev
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)          # noqa: E731
        self.offs = z(5 * (s.B + 1), i32).view(5, s.B + 1)
        self.offs_host = PinnedRing((5, s.B + 1), i32)
        self.pair_cnt, self.pair_off = z(s.B, i32), z(s.B + 1, i32)
        slack = s.N * s.N                                               # a molecule that straddles the pair capacity
        self.t = {'cx': z(s.V, i32), 'cxp': z(s.V, i32), 'cxpos': z(s.V, i32),
                  'row_ptr': z(s.V + 1, i32), 'col': z(max(s.E, 1), i32), 'cls': z(max(s.E, 1), u8),
                  'in_row_ptr': z(s.V + 1, i32), 'in_col': z(max(s.Ein, 1), i32), 'in_cls': z(max(s.Ein, 1), u8),
                  'pair_src': z(s.P + slack, i32), 'pair_dst': z(s.P + slack, i32), 'pair_graph': z(s.P + slack, i32),
                  'golden32': z(s.P + slack, i32)}
        self.pieces = torch.zeros(s.B, s.S, dtype=torch.int64, device=dev)
        self.props = torch.zeros(s.B, 3, dtype=torch.float32, device=dev)
        self.num_bond_type = num_bond_type
        node_off = self.offs[0]
        segments = ops.Segments(node_off[:-1], node_off[1:], False)      # padding rows belong to no molecule
        compact = CompactBatch.__new__(CompactBatch)
        compact.cx, compact.cxp, compact.cxpos = self.t['cx'], self.t['cxp'], self.t['cxpos']
        compact.node_off, compact.pair_off = node_off, self.pair_off
        compact.csr = ops.Csr(self.t['row_ptr'], self.t['col'], self.t['cls'])
        compact.csr_in = ops.Csr(self.t['in_row_ptr'], self.t['in_col'], self.t['in_cls'])
        compact.segments = segments
        pairs = ops.PairList(self.t['pair_src'][:s.P], self.t['pair_dst'][:s.P], self.t['pair_graph'][:s.P], self.pair_off,
                             segments)
        pairs.live = self.pair_off[s.B:s.B + 1]                          # pairs of this batch, read on the device
        compact._pairs, compact._golden, compact._pending = pairs, self.t['golden32'][:s.P], None
        compact._pair_bufs = None
        compact.num_graphs = s.B
        self.batch = CollatedBatch()
        self.batch.compact = compact
        self.batch.pieces, self.batch.props, self.batch.batch_size = self.pieces, self.props, s.B


def collate_static(dpacked, ids, static, uniforms=None, atom_pad=10, piece_pad=300, keep_prob=KEEP_PROB, generator=None):
    """Collate molecules `ids` into the fixed-size buffers of `static` (StaticBatch).  Returns its
    CollatedBatch (compact layout only), or None when the batch does not fit the static sizes.

    Padding: atom rows beyond the batch's own V have no bonds (their CSR rows are empty) and
    belong to no molecule, pair rows beyond the batch's own P lie behind `pairs.live` - so they
    feed no sum, receive a zero gradient and contribute nothing to any weight gradient.  A batch
    that would keep more than static.P pairs keeps the first static.P (by molecule order)."""
    s = static.shapes
    p = plan(dpacked.host, ids)
    if p['B'] != s.B:
        raise ValueError(f'a static batch holds {s.B} molecules, {p["B"]} given')
    if p['V'] > s.V or p['E'] > s.E or p['Ein'] > s.Ein or p['R'] > s.R or p['S'] > s.S or p['N'] > s.N:
        return None
    host = static.offs_host.next()
    host.copy_(torch.from_numpy(p['offs']))
    static.offs.copy_(host, non_blocking=True)
    static.offs_host.sent()
    node_off, edge_off, inedge_off, rng_off, mol_ids = static.offs
    dev = static.device
    if uniforms is None:
        uniforms = torch.rand(max(s.R, 1), dtype=torch.float64, device=dev, generator=generator)
    elif uniforms.numel() < p['R']:
        raise ValueError(f'{p["R"]} uniforms needed, {uniforms.numel()} given')
    hip.collate_count(s.B, s.N, mol_ids, dpacked, rng_off, uniforms, keep_prob, static.pair_cnt, static.pair_off)
    static.pair_off.clamp_(max=s.P)
    t = static.t
    t['row_ptr'].fill_(p['E'])            # padding rows: empty CSR rows behind the last edge
    t['in_row_ptr'].fill_(p['Ein'])
    hip.collate_fill(s.B, s.N, s.S, mol_ids, dpacked,
                     {'node_off': node_off, 'edge_off': edge_off, 'inedge_off': inedge_off, 'rng_off': rng_off,
                      'pair_off': static.pair_off},
                     p['E'], uniforms, keep_prob, atom_pad, piece_pad, static.num_bond_type,
                     {'pieces': static.pieces, 'props': static.props}, t)
    static.batch.num_uniforms = p['R']
    return static.batch
