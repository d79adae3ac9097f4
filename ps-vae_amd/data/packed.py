"""Packed molecules: a flat, ragged, int32 re-encoding of the per-molecule dict that
BPEMolDataset.process_step1 returns (reference: src/data/bpe_dataset.py:47-54).

The reference keeps a Python list of dicts ('x', 'edge_index', 'edge_attr', 'props',
'pieces', 'groups') and expands every molecule into dense n x n Python lists in
process_step2 (:56-84) on each access.  The collation kernels read this packed form
straight from HBM instead:

    atom_off  [M+1]  atoms of molecule i are atom_off[i] .. atom_off[i+1]
    atom_type [A]    'x'
    atom_pos  [A]    position of the group ('groups') that holds the atom, 0 = in no group.
                     Groups partition the atoms (src/data/mol_bpe.py:192-222), so this is
                     the whole content of 'groups'; x_pos = atom_pos and
                     x_pieces = pieces[atom_pos] (bpe_dataset.py:72-74).
    bond_off  [M+1], bond_a/bond_b/bond_cls [Bd]   undirected bonds, molecule-local atom
                     ids, duplicates resolved the way the dense assignment at :61-63 does
                     (the last one wins), class-0 (None) entries dropped.
    piece_off [M+1], piece_id [S]   'pieces' (with the '<s>' id at both ends)
    props     [M,3]  float32
    n_edge / n_inpiece / n_rng [M]  per-molecule totals the batch planner prefix-sums:
                     directed edges, directed in-piece edges, and the number of non-bonded
                     candidate pairs (= uniforms drawn at bpe_dataset.py:129).
"""
import numpy as np
import torch

_INT_FIELDS = ('atom_off', 'atom_type', 'atom_pos', 'bond_off', 'bond_a', 'bond_b', 'bond_cls',
               'piece_off', 'piece_id', 'n_edge', 'n_inpiece', 'n_rng')


def load_step1_pickle(path):
    """The reference's dataset cache: a torch.save'd list of process_step1 dicts
    ('processed_data.pkl', src/data/bpe_dataset.py:23-26, 170).  Read with
    weights_only=True, so a file holding anything but lists / dicts / numbers / tensors is
    refused instead of executed."""
    data = torch.load(path, map_location='cpu', weights_only=True)
    if not isinstance(data, (list, tuple)) or (len(data) and not isinstance(data[0], dict)):
        raise ValueError(f'{path}: expected a list of process_step1 dicts')
    keys = ('x', 'edge_index', 'edge_attr', 'props', 'pieces', 'groups')
    for i, d in enumerate(data):
        missing = [k for k in keys if k not in d]
        if missing:
            raise ValueError(f'{path}: molecule {i} lacks {missing}')
    return [{k: (d[k].tolist() if torch.is_tensor(d[k]) else d[k]) for k in keys} for d in data]


class PackedMolecules:
    def __init__(self, **arrays):
        for k in _INT_FIELDS:
            setattr(self, k, np.ascontiguousarray(arrays[k], dtype=np.int32))
        self.props = np.ascontiguousarray(arrays['props'], dtype=np.float32).reshape(len(self), -1)

    def __len__(self):
        return len(self.atom_off) - 1

    @property
    def n_atoms(self):
        return np.diff(self.atom_off)

    @property
    def n_pieces(self):
        return np.diff(self.piece_off)

    @classmethod
    def from_step1(cls, data_list):
        """Pack a list of process_step1 dicts.  Raises ValueError where the dict cannot be
        represented (an atom listed in two groups, an atom id out of range)."""
        atom_off, bond_off, piece_off = [0], [0], [0]
        atom_type, atom_pos, bond_a, bond_b, bond_cls, piece_id = [], [], [], [], [], []
        props, n_edge, n_inpiece, n_rng = [], [], [], []
        for idx, d in enumerate(data_list):
            n = len(d['x'])
            pos = [0] * n
            if len(d['groups']) != len(d['pieces']):
                raise ValueError(f'molecule {idx}: {len(d["groups"])} groups for {len(d["pieces"])} pieces')
            for p, group in enumerate(d['groups']):
                if p == 0 and len(group):
                    raise ValueError(f'molecule {idx}: group 0 (the <s> slot) must be empty')
                for aid in group:
                    if not 0 <= aid < n:
                        raise ValueError(f'molecule {idx}: atom id {aid} out of range')
                    if pos[aid]:
                        raise ValueError(f'molecule {idx}: atom {aid} is in two groups')
                    pos[aid] = p
            bonds = {}
            for (a, b), c in zip(d['edge_index'], d['edge_attr']):
                if not (0 <= a < n and 0 <= b < n):
                    raise ValueError(f'molecule {idx}: bond ({a},{b}) out of range')
                bonds[(min(a, b), max(a, b))] = int(c)
            e = e_in = 0
            for (a, b), c in bonds.items():
                if c == 0:
                    continue
                bond_a.append(a), bond_b.append(b), bond_cls.append(c)
                mult = 1 if a == b else 2
                e += mult
                if pos[a] and pos[a] == pos[b]:
                    e_in += mult
            sizes = np.bincount(np.asarray(pos, dtype=np.int64), minlength=1)[1:]
            same_cells = int((sizes.astype(np.int64) ** 2).sum())
            atom_type += list(d['x'])
            atom_pos += pos
            piece_id += list(d['pieces'])
            props.append(list(d['props']))
            n_edge.append(e), n_inpiece.append(e_in)
            n_rng.append(n * n - same_cells - (e - e_in))
            atom_off.append(len(atom_type)), bond_off.append(len(bond_a)), piece_off.append(len(piece_id))
        return cls(atom_off=atom_off, atom_type=atom_type, atom_pos=atom_pos, bond_off=bond_off,
                   bond_a=bond_a, bond_b=bond_b, bond_cls=bond_cls, piece_off=piece_off,
                   piece_id=piece_id, props=np.asarray(props, dtype=np.float32).reshape(len(props), -1),
                   n_edge=n_edge, n_inpiece=n_inpiece, n_rng=n_rng)

    def to_step1(self, i):
        """Inverse of from_step1 for molecule i (group member order is ascending atom id)."""
        a0, a1 = self.atom_off[i], self.atom_off[i + 1]
        b0, b1 = self.bond_off[i], self.bond_off[i + 1]
        p0, p1 = self.piece_off[i], self.piece_off[i + 1]
        groups = [[] for _ in range(p1 - p0)]
        for aid, p in enumerate(self.atom_pos[a0:a1]):
            if p:
                groups[p].append(aid)
        return {'x': self.atom_type[a0:a1].tolist(),
                'edge_index': [[int(a), int(b)] for a, b in zip(self.bond_a[b0:b1], self.bond_b[b0:b1])],
                'edge_attr': self.bond_cls[b0:b1].tolist(),
                'props': self.props[i].tolist(),
                'pieces': self.piece_id[p0:p1].tolist(),
                'groups': groups}

    def select(self, ids):
        """A new PackedMolecules holding molecules `ids` in that order."""
        ids = np.asarray(ids, dtype=np.int64)

        def ragged(off, *vals):
            lens = (off[ids + 1] - off[ids]).astype(np.int64)
            new_off = np.concatenate([[0], np.cumsum(lens)])
            take = np.repeat(off[ids].astype(np.int64) - new_off[:-1], lens) + np.arange(new_off[-1])
            return (new_off,) + tuple(v[take] for v in vals)

        ao, at, ap = ragged(self.atom_off, self.atom_type, self.atom_pos)
        bo, ba, bb, bc = ragged(self.bond_off, self.bond_a, self.bond_b, self.bond_cls)
        po, pi = ragged(self.piece_off, self.piece_id)
        return PackedMolecules(atom_off=ao, atom_type=at, atom_pos=ap, bond_off=bo, bond_a=ba,
                               bond_b=bb, bond_cls=bc, piece_off=po, piece_id=pi,
                               props=self.props[ids], n_edge=self.n_edge[ids],
                               n_inpiece=self.n_inpiece[ids], n_rng=self.n_rng[ids])

    def save(self, path):
        np.savez_compressed(path, props=self.props, **{k: getattr(self, k) for k in _INT_FIELDS})

    @classmethod
    def load(cls, path):
        with np.load(path, allow_pickle=False) as f:
            return cls(**{k: f[k] for k in f.files})

    def to_device(self, device):
        return DevicePackedMolecules(self, device)


class DevicePackedMolecules:
    """The packed arrays as int32 / float32 device tensors plus the host copies of the
    per-molecule counts the batch planner needs."""

    def __init__(self, packed, device):
        self.host = packed
        self.device = torch.device(device)
        for k in _INT_FIELDS:
            setattr(self, k, torch.from_numpy(getattr(packed, k)).to(self.device))
        self.props = torch.from_numpy(packed.props).to(self.device)

    def __len__(self):
        return len(self.host)
