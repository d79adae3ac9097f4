"""TEST INFRASTRUCTURE ONLY (CPU restatement, never imported by the product path).

The adjacency-only part of VAEPieceDecoder.inference: valence_check and cycle_check
(src/utils/chem_utils.py:447-483) and the greedy bond loop with block merging and
largest-block selection (src/modules/vae_piece_decoder.py:245-270), on plain Python lists
instead of an RDKit RWMol (only GetNeighbors() of that object is used by the reference's
checks).  valence_check / cycle_check are pinned by tests/golden/assemble_checks.npz, which
tests/golden/make_golden.py produced by calling the reference's own functions; the loop
around them is restated from the lines cited and is parity unpinned beyond those two."""
from collections import deque

MAX_VALENCE = {'B': 3, 'Br': 1, 'C': 4, 'Cl': 1, 'F': 1, 'I': 1, 'N': 3, 'O': 2, 'P': 5, 'S': 6}   # chem_utils.py:14


def valence_check(aid1, aid2, edges1, edges2, new_edge, vocab):
    """chem_utils.py:471-483 with c1 = c2 = 0 (the call at vae_piece_decoder.py:250-251)."""
    new_valence = vocab.bond_idx_to_valence(new_edge)
    if new_valence == -1:
        return False
    atom1, atom2 = vocab.idx_to_atom(aid1), vocab.idx_to_atom(aid2)
    a1_val = sum(vocab.bond_idx_to_valence(e) for e in edges1)
    a2_val = sum(vocab.bond_idx_to_valence(e) for e in edges2)
    if (atom1 == 'S' and a1_val == 2) or (atom2 == 'S' and a2_val == 2):
        return False
    return a1_val + new_valence <= MAX_VALENCE[atom1] and a2_val + new_valence <= MAX_VALENCE[atom2]


def shortest_path_len(i, j, adj):
    """chem_utils.py:447-463: atoms on the shortest path from i to j (2 if bonded), None if
    j cannot be reached.  adj[a] = list of neighbours."""
    queue = deque([(i, 1)])
    visited = {i}
    while queue:
        atom, dist = queue.popleft()
        for idx in adj[atom]:
            if idx == j:
                return dist + 1
            if idx not in visited:
                visited.add(idx)
                queue.append((idx, dist + 1))
    return None


def cycle_check(i, j, adj):
    """chem_utils.py:466-468."""
    cycle_len = shortest_path_len(i, j, adj)
    return cycle_len is None or (4 < cycle_len < 7)


def assemble(atoms, bonds, groups, cand_pairs, cand_type, sorted_idx, vocab):
    """vae_piece_decoder.py:245-270.  atoms: atom vocabulary ids; bonds: existing
    (begin, end, bond idx) inside the fragments; groups: atom ids per fragment; cand_pairs /
    cand_type: every candidate; sorted_idx: the order to try them in.
    Returns (accepted flags per entry of sorted_idx, final block id per atom, kept atom flags)."""
    n = len(atoms)
    adj = [[] for _ in range(n)]
    edge_sets = [[] for _ in range(n)]
    for b, e, t in bonds:
        adj[b].append(e), adj[e].append(b)
        edge_sets[b].append(t), edge_sets[e].append(t)
    aid2bid = {}
    for gi, group in enumerate(groups):
        for aid in group:
            aid2bid[aid] = gi
    block_atom_cnt = [len(g) for g in groups]
    accepted = []
    for i in sorted_idx:
        begin, end = int(cand_pairs[i][0]), int(cand_pairs[i][1])
        bond_type = int(cand_type[i])
        ok = valence_check(atoms[begin], atoms[end], edge_sets[begin], edge_sets[end], bond_type, vocab) and \
            cycle_check(begin, end, adj)
        accepted.append(ok)
        if ok:
            adj[begin].append(end), adj[end].append(begin)
            edge_sets[begin].append(bond_type), edge_sets[end].append(bond_type)
            bid1, bid2 = aid2bid[begin], aid2bid[end]
            if bid1 != bid2:
                for aid in aid2bid:
                    if aid2bid[aid] == bid1:
                        aid2bid[aid] = bid2
                block_atom_cnt[bid2] += block_atom_cnt[bid1]
    bid = max(range(len(block_atom_cnt)), key=lambda i: block_atom_cnt[i])
    return accepted, [aid2bid[a] for a in range(n)], [aid2bid[a] == bid for a in range(n)]
