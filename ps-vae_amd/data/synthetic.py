"""Synthetic ZINC-shaped / QM9-shaped molecules in the reference's process_step1 dict
format (src/data/bpe_dataset.py:47-54).  RDKit, the 300-piece vocabulary and
data/zinc250k/train.txt are not available offline, so throughput is measured on graphs with
the commonly cited ZINC250K statistics: ~23 heavy atoms (<= 38), ~25 bonds, C-dominated
atom mix, kekulised bond classes, ~n/4 connected principal-subgraph groups."""
import numpy as np

ATOM_P = np.array([0.001, 0.002, 0.74, 0.007, 0.014, 0.001, 0.12, 0.10, 0.001, 0.015])
ATOM_P = ATOM_P / ATOM_P.sum()   # order of MAX_VALENCE keys, src/utils/chem_utils.py:14
BOND_P = np.array([0.76, 0.23, 0.01])


def random_molecule(rng, n, num_pieces=300, end_idx=301, max_degree=4):
    deg = np.zeros(n, dtype=np.int64)
    bonds = set()
    order = rng.permutation(n)
    for k in range(1, n):                         # random spanning tree
        a = int(order[k])
        cand = [int(b) for b in order[:k] if deg[b] < max_degree]
        b = cand[int(rng.integers(len(cand)))] if cand else int(order[int(rng.integers(k))])
        bonds.add((min(a, b), max(a, b)))
        deg[a] += 1
        deg[b] += 1
    extra = int(rng.integers(1, 4)) if n >= 5 else 0   # ring closures: n + 2 bonds on average
    tries = 0
    while extra and tries < 50:
        tries += 1
        a, b = (int(v) for v in rng.integers(n, size=2))
        if a == b or (min(a, b), max(a, b)) in bonds or deg[a] >= max_degree or deg[b] >= max_degree:
            continue
        bonds.add((min(a, b), max(a, b)))
        deg[a] += 1
        deg[b] += 1
        extra -= 1
    bonds = sorted(bonds)
    adj = [[] for _ in range(n)]
    for a, b in bonds:
        adj[a].append(b)
        adj[b].append(a)
    k = max(1, int(round(n / 4)))                 # connected groups grown from k seeds
    owner = -np.ones(n, dtype=np.int64)
    seeds = rng.choice(n, size=min(k, n), replace=False)
    frontier = []
    for g, s in enumerate(seeds):
        owner[s] = g
        frontier.append(int(s))
    while frontier:
        i = int(rng.integers(len(frontier)))
        a = frontier[i]
        free = [b for b in adj[a] if owner[b] < 0]
        if not free:
            frontier.pop(i)
            continue
        b = free[int(rng.integers(len(free)))]
        owner[b] = owner[a]
        frontier.append(b)
    groups = [[int(a) for a in np.nonzero(owner == g)[0]] for g in range(len(seeds))]
    groups = [groups[i] for i in rng.permutation(len(groups))]   # order shuffled, mol_bpe.py:214
    piece_ids = [int(v) for v in rng.integers(num_pieces, size=len(groups))]
    return {
        'x': [int(v) for v in rng.choice(len(ATOM_P), size=n, p=ATOM_P)],
        'edge_index': [[a, b] for a, b in bonds],
        'edge_attr': [int(v) + 1 for v in rng.choice(3, size=len(bonds), p=BOND_P)],
        'props': [float(v) for v in rng.random(3)],
        'pieces': [end_idx] + piece_ids + [end_idx],
        'groups': [[]] + groups + [[]],
    }


def synthetic_step1(num, seed=2021, shape='zinc', num_pieces=300):
    """`num` molecules; shape 'zinc': n ~ clip(round(N(23.2, 4.5)), 6, 38); 'qm9': n in [1, 9]."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num):
        if shape == 'zinc':
            n = int(np.clip(round(rng.normal(23.2, 4.5)), 6, 38))
        elif shape == 'qm9':
            n = int(rng.integers(1, 10))
        else:
            raise ValueError(shape)
        out.append(random_molecule(rng, n, num_pieces=num_pieces, end_idx=num_pieces + 1))
    return out
