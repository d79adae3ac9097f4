"""CPU restatement (pure Python / numpy) of PS-VAE's batch collation, the integer half of
the training hot path.  TEST INFRASTRUCTURE ONLY (see oracle/model_ref.py header).

Restates src/data/bpe_dataset.py:56-84 (process_step2) and :86-155 (process_step3) of the
reference.  The reference draws `random() < 0.022` (Python's MT19937 53-bit doubles) for
every non-bonded candidate pair, in row-major (molecule, m, n) order, short-circuited so
that bonded pairs draw nothing (:127-130); here the uniform stream is an explicit array
consumed in exactly that order, so results are bit-exact for a given stream.

Pinned by tests/golden/collate_*.npz (outputs of the reference's own process_step2/3 run
under random.seed with the drawn uniforms recorded).

Besides the reference's batch dict this module also derives the flat "compact" layout the
HIP kernels consume (real atoms only, CSR adjacency, pair lists); its definition is in
`compact_from_reference`.
"""
import numpy as np

# Vocabulary constants: src/utils/chem_utils.py:14-15 (10 atom symbols, bond classes
# None/SINGLE/DOUBLE/TRIPLE), src/data/mol_bpe.py:167-173 ('<pad>' then '<s>' appended
# after the pieces; atom '<pad>' appended after the atoms).
NONE_BOND = 0
NUM_BOND_TYPE = 4
ATOM_PAD = 10
KEEP_PROB = 0.022   # src/data/bpe_dataset.py:129


def process_step2(data, none_bond=NONE_BOND):
    """src/data/bpe_dataset.py:56-84."""
    x = data['x']
    n = len(x)
    ad_mat = [[none_bond] * n for _ in range(n)]
    for (a, b), attr in zip(data['edge_index'], data['edge_attr']):
        ad_mat[a][b] = ad_mat[b][a] = attr
    x_pieces, x_pos = [0] * n, [0] * n
    edge_select = [[1] * n for _ in range(n)]
    for pos, group in enumerate(data['groups']):
        for i in range(len(group)):
            for j in range(i, len(group)):
                edge_select[group[i]][group[j]] = edge_select[group[j]][group[i]] = 0
        for aid in group:
            x_pieces[aid] = data['pieces'][pos]
            x_pos[aid] = pos
    return {'x': x, 'ad_mat': ad_mat, 'props': data['props'], 'pieces': data['pieces'],
            'x_pieces': x_pieces, 'x_pos': x_pos, 'edge_select': edge_select}


def process_step3(data_list, uniforms, piece_pad, atom_pad=ATOM_PAD, none_bond=NONE_BOND,
                  num_bond_type=NUM_BOND_TYPE, keep_prob=KEEP_PROB):
    """src/data/bpe_dataset.py:86-155 on numpy arrays.  `uniforms` replaces the calls of
    random() at :129.  Returns (batch dict, number of uniforms consumed)."""
    B = len(data_list)
    lens = [len(d['x']) for d in data_list]
    N = max(lens) if B else 0
    S = max(len(d['pieces']) for d in data_list) if B else 0
    x = np.full((B, N), atom_pad, dtype=np.int64)
    x_pieces = np.full((B, N), piece_pad, dtype=np.int64)
    x_pos = np.zeros((B, N), dtype=np.int64)
    atom_mask = np.zeros((B, N), dtype=bool)
    pieces = np.full((B, S), piece_pad, dtype=np.int64)
    edge_select = np.zeros((B, N, N), dtype=bool)
    edge_index, edge_attr, golden_edge, in_piece_edge_idx = [], [], [], []
    u = 0
    for i, d in enumerate(data_list):
        n = lens[i]
        x[i, :n] = d['x']
        x_pieces[i, :n] = d['x_pieces']
        x_pos[i, :n] = d['x_pos']
        atom_mask[i, n:] = True                      # :101-103 (cumsum trick)
        pieces[i, :len(d['pieces'])] = d['pieces']
        off = N * i
        for m in range(n):
            for k in range(n):
                val, attr = d['edge_select'][m][k], d['ad_mat'][m][k]
                if attr != none_bond:                # :119-125
                    edge_index.append((m + off, k + off))
                    edge_attr.append(attr)
                    if val == 0:
                        in_piece_edge_idx.append(len(edge_index) - 1)
                if val == 1:                         # :127-132
                    keep = attr != none_bond
                    if not keep:
                        keep = uniforms[u] < keep_prob
                        u += 1
                    if keep:
                        golden_edge.append(attr)
                        edge_select[i, m, k] = True
    ei = np.asarray(edge_index, dtype=np.int64).reshape(-1, 2).T.copy()
    ea = np.asarray(edge_attr, dtype=np.int64)
    one_hot = np.zeros((len(ea), num_bond_type), dtype=np.int64)
    one_hot[np.arange(len(ea)), ea] = 1
    batch = {
        'x': x, 'x_pieces': x_pieces, 'x_pos': x_pos, 'atom_mask': atom_mask, 'pieces': pieces,
        'edge_index': ei, 'edge_attr': one_hot, 'edge_select': edge_select,
        'golden_edge': np.asarray(golden_edge, dtype=np.int64),
        'in_piece_edge_idx': np.asarray(in_piece_edge_idx, dtype=np.int64),
        'props': np.asarray([d['props'] for d in data_list], dtype=np.float32).reshape(B, -1),
    }
    return batch, u


def collate(step1_list, uniforms, piece_pad, **kw):
    """process_step2 on every molecule, then process_step3 (the DataLoader's
    __getitem__ + collate_fn, src/data/bpe_dataset.py:173-181)."""
    return process_step3([process_step2(d) for d in step1_list], uniforms, piece_pad, **kw)


def compact_from_reference(batch):
    """Derive the compact layout from a reference-format batch (pure index bookkeeping):

      node_off [B+1]   prefix sum of real-atom counts; compact id of atom (b, m) = node_off[b] + m
      cx, cxp, cxpos   [V] atom / piece / position ids of the real atoms
      row_ptr [V+1], col [E], cls [E]          adjacency in edge_index order (rows = source
                       atom; the graph is symmetric, so row i also lists the edges INTO i)
      in_row_ptr, in_col, in_cls               the same for the in-piece edge subset
      pair_src, pair_dst, pair_graph, golden [P]   selected candidate pairs, row-major
      pair_ptr [B+1]   pairs of graph b are pair_ptr[b] .. pair_ptr[b+1]
    """
    B, N = batch['x'].shape
    real = ~batch['atom_mask']
    n_atoms = real.sum(axis=1).astype(np.int64)
    node_off = np.concatenate([[0], np.cumsum(n_atoms)]).astype(np.int32)
    V = int(node_off[-1])

    def to_compact(padded_id):
        b = padded_id // N
        return (node_off[b] + (padded_id - b * N)).astype(np.int32)

    out = {'node_off': node_off,
           'cx': batch['x'][real].astype(np.int32),
           'cxp': batch['x_pieces'][real].astype(np.int32),
           'cxpos': batch['x_pos'][real].astype(np.int32)}

    def csr(src_padded, dst_padded, cls):
        src = to_compact(src_padded) if len(src_padded) else np.zeros(0, np.int32)
        dst = to_compact(dst_padded) if len(dst_padded) else np.zeros(0, np.int32)
        assert np.all(np.diff(src) >= 0), 'edge_index row 0 must be ascending'
        row_ptr = np.zeros(V + 1, dtype=np.int32)
        np.add.at(row_ptr, src + 1, 1)
        return np.cumsum(row_ptr).astype(np.int32), dst, cls.astype(np.uint8)

    ei = batch['edge_index']
    cls = batch['edge_attr'].argmax(axis=1) if len(batch['edge_attr']) else np.zeros(0, np.int64)
    out['row_ptr'], out['col'], out['cls'] = csr(ei[0], ei[1], cls)
    idx = batch['in_piece_edge_idx']
    out['in_row_ptr'], out['in_col'], out['in_cls'] = csr(ei[0][idx], ei[1][idx], cls[idx])
    b, m, n = np.nonzero(batch['edge_select'])
    out['pair_src'] = (node_off[b] + m).astype(np.int32)
    out['pair_dst'] = (node_off[b] + n).astype(np.int32)
    out['pair_graph'] = b.astype(np.int32)
    out['golden'] = batch['golden_edge'].astype(np.int32)
    pair_ptr = np.zeros(B + 1, dtype=np.int32)
    np.add.at(pair_ptr, b + 1, 1)
    out['pair_ptr'] = np.cumsum(pair_ptr).astype(np.int32)
    return out
