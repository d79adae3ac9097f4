"""Per-vocabulary fragment table: every piece of the vocabulary as atoms + bonds.

`VAEPieceDecoder.inference` turns each sampled piece id into atoms and in-piece bonds by
parsing the piece's SMILES with RDKit (src/modules/vae_piece_decoder.py:160-200:
smiles2molecule(smi, kekulize=True), then GetAtoms / GetBonds).  The vocabulary is fixed, so
that work is done ONCE per vocabulary here and batched generation only gathers from flat
arrays:

    atom_off [P+1], atom_type [A]            chemical-vocabulary index of every atom
    edge_off [P+1], edge_src/edge_dst/edge_cls [2*bonds]
                                             directed in-piece edges with piece-local atom
                                             ids, sorted by source (the CSR order)

Pieces are written by the reference from kekulised molecules (src/data/mol_bpe.py:20-71,
cnt_atom at :147 only counts upper-case element letters), so their SMILES use explicit
single / double / triple bonds and upper-case atoms; `parse_piece_smiles` reads exactly that
subset (organic-subset and bracket atoms, branches, ring closures incl. %nn, '.', stereo marks
ignored like Chem.RemoveStereochemistry) with atoms numbered in order of appearance, as any
SMILES reader numbers them.  Aromatic (lower-case) atoms are refused: kekulising them is
RDKit's job; a table for such a vocabulary can be produced with RDKit once and loaded with
FragmentTable.load.  The ORDER of the bonds inside a piece (chain bonds as they are read, ring
closures when they close) only decides the order in which a node's messages are summed.
"""
import numpy as np

from .chem_vocab import MAX_VALENCE

_BOND = {'-': 1, '=': 2, '#': 3, '/': 1, '\\': 1}
_TWO_LETTER = ('Cl', 'Br')


def parse_piece_smiles(smi):
    """(symbols, bonds): element symbol of every atom in order of appearance and
    [(begin, end, order)] with order in 1..3."""
    symbols, bonds = [], []
    stack, prev, pending = [], None, None      # branch stack, previous atom, bond symbol waiting
    rings = {}                                  # ring label -> (atom, bond order or None)
    i, n = 0, len(smi)

    def add_atom(sym):
        nonlocal prev, pending
        if sym not in MAX_VALENCE:
            raise ValueError(f'piece {smi!r}: element {sym!r} is not in the chemical vocabulary')
        symbols.append(sym)
        cur = len(symbols) - 1
        if prev is not None:
            bonds.append((prev, cur, pending or 1))
        prev, pending = cur, None

    def ring(label):
        nonlocal pending
        if prev is None:
            raise ValueError(f'piece {smi!r}: ring closure before any atom')
        if label in rings:
            other, order = rings.pop(label)
            bonds.append((other, prev, pending or order or 1))
        else:
            rings[label] = (prev, pending)
        pending = None

    while i < n:
        c = smi[i]
        if c == '[':
            j = smi.find(']', i)
            if j < 0:
                raise ValueError(f'piece {smi!r}: unclosed bracket atom')
            body = smi[i + 1:j].lstrip('0123456789')                 # isotope
            sym = body[:2] if body[:2] in MAX_VALENCE else body[:1]
            if sym.islower():
                raise ValueError(f'piece {smi!r}: aromatic atom {sym!r} (kekulise the vocabulary first)')
            add_atom(sym)
            i = j + 1
        elif smi[i:i + 2] in _TWO_LETTER:
            add_atom(smi[i:i + 2])
            i += 2
        elif c.isalpha():
            if c.islower():
                raise ValueError(f'piece {smi!r}: aromatic atom {c!r} (kekulise the vocabulary first)')
            add_atom(c)
            i += 1
        elif c in _BOND:
            pending = _BOND[c]
            i += 1
        elif c == ':':
            raise ValueError(f'piece {smi!r}: aromatic bond (kekulise the vocabulary first)')
        elif c == '(':
            stack.append(prev)
            i += 1
        elif c == ')':
            if not stack:
                raise ValueError(f'piece {smi!r}: unbalanced parenthesis')
            prev = stack.pop()
            i += 1
        elif c.isdigit():
            ring(c)
            i += 1
        elif c == '%':
            ring(smi[i + 1:i + 3])
            i += 3
        elif c == '.':
            prev, pending = None, None
            i += 1
        else:
            raise ValueError(f'piece {smi!r}: unexpected character {c!r}')
    if rings or stack:
        raise ValueError(f'piece {smi!r}: unclosed ring or branch')
    return symbols, bonds


class FragmentTable:
    def __init__(self, atom_off, atom_type, edge_off, edge_src, edge_dst, edge_cls):
        self.atom_off = np.ascontiguousarray(atom_off, dtype=np.int32)
        self.atom_type = np.ascontiguousarray(atom_type, dtype=np.int32)
        self.edge_off = np.ascontiguousarray(edge_off, dtype=np.int32)
        self.edge_src = np.ascontiguousarray(edge_src, dtype=np.int32)
        self.edge_dst = np.ascontiguousarray(edge_dst, dtype=np.int32)
        self.edge_cls = np.ascontiguousarray(edge_cls, dtype=np.uint8)
        self._device = {}

    def __len__(self):
        return len(self.atom_off) - 1

    @property
    def atom_counts(self):
        return np.diff(self.atom_off)

    @classmethod
    def from_pieces(cls, pieces, chem_vocab):
        """pieces: per piece id either a SMILES string or an (symbols, bonds) pair; special
        tokens ('<pad>', '<s>') become empty fragments."""
        atom_off, edge_off = [0], [0]
        atom_type, src, dst, klass = [], [], [], []
        for piece in pieces:
            if isinstance(piece, str):
                symbols, bonds = ([], []) if piece.startswith('<') else parse_piece_smiles(piece)
            else:
                symbols, bonds = piece
            atom_type += [chem_vocab.atom_to_idx(s) for s in symbols]
            directed = [(a, b, c) for a, b, c in bonds] + [(b, a, c) for a, b, c in bonds]
            directed.sort(key=lambda e: e[0])                  # stable: CSR by source, bond order kept
            for a, b, c in directed:
                if not (0 <= a < len(symbols) and 0 <= b < len(symbols) and 1 <= c <= 3):
                    raise ValueError(f'bad bond {(a, b, c)} in piece {piece!r}')
                src.append(a), dst.append(b), klass.append(c)
            atom_off.append(len(atom_type)), edge_off.append(len(src))
        return cls(atom_off, atom_type, edge_off, src, dst, klass)

    @classmethod
    def from_tokenizer(cls, tokenizer):
        return cls.from_pieces([tokenizer.idx_to_piece(i) for i in range(tokenizer.num_piece_type())],
                               tokenizer.chem_vocab)

    def save(self, path):
        np.savez_compressed(path, atom_off=self.atom_off, atom_type=self.atom_type, edge_off=self.edge_off,
                            edge_src=self.edge_src, edge_dst=self.edge_dst, edge_cls=self.edge_cls)

    @classmethod
    def load(cls, path):
        with np.load(path, allow_pickle=False) as f:
            return cls(**{k: f[k] for k in f.files})

    def on(self, device):
        """The arrays as device tensors (cached per device)."""
        import torch
        key = str(device)
        if key not in self._device:
            self._device[key] = {k: torch.from_numpy(getattr(self, k)).to(device)
                                 for k in ('atom_off', 'atom_type', 'edge_off', 'edge_src', 'edge_dst', 'edge_cls')}
        return self._device[key]
