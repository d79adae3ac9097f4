"""The reference's Tokenizer (src/data/mol_bpe.py:156-249): vocabulary file, special-token layout,
id lookups, and tokenising a molecule (the greedy principal-subgraph merge, :175-222, in
data.mol_bpe on the plain graphs of utils.molgraph instead of RDKit molecules)."""
import numpy as np

from .chem_vocab import GeneralVocab, cnt_atom


def synthetic_pieces(num, seed=0):
    """`num` distinct kekulised piece SMILES of 1..8 atoms (chains with a branch, a double bond
    or a 5/6-ring here and there, valences respected), largest first like a mined vocabulary
    (src/data/mol_bpe.py:148-149).  Stand-in for the 300-piece vocabulary file, which is not
    available offline: the pieces are chemically plausible but not mined from data."""
    rng = np.random.default_rng(seed)
    heavy = ['C', 'N', 'O', 'S']
    cap = {'C': 4, 'N': 3, 'O': 2, 'S': 2, 'F': 1, 'Cl': 1, 'Br': 1, 'I': 1}
    out = ['C', 'N', 'O', 'S', 'F', 'Cl', 'Br', 'I'][:num]
    seen = set(out)
    while len(out) < num:
        n = int(rng.integers(2, 9))
        atoms = [str(rng.choice(heavy, p=[0.74, 0.13, 0.10, 0.03])) for _ in range(n)]
        if rng.random() < 0.15:
            atoms[-1] = str(rng.choice(['F', 'Cl', 'Br', 'I']))
        ring = n >= 5 and rng.random() < 0.35 and cap[atoms[-1]] > 1
        used = [0] * n
        bond = [1] * (n - 1)
        for i in range(n - 1):
            used[i] += 1
            used[i + 1] += 1
        if ring:
            used[0] += 1
            used[n - 1] += 1
        for i in range(n - 1):
            if rng.random() < 0.2 and used[i] < cap[atoms[i]] and used[i + 1] < cap[atoms[i + 1]]:
                bond[i] = 2
                used[i] += 1
                used[i + 1] += 1
        if any(u > cap[a] for u, a in zip(used, atoms)):
            continue
        branch = int(rng.integers(1, n - 1)) if (n >= 4 and not ring and rng.random() < 0.3) else -1
        smi = atoms[0] + ('1' if ring else '')
        for i in range(1, n):
            b = '=' if bond[i - 1] == 2 else ''
            if i == branch and used[i - 1] <= cap[atoms[i - 1]]:
                smi += '(' + b + atoms[i] + ')'
                # the next atom now bonds to atom i-1 instead of atom i: same degree counts
                if i + 1 < n:
                    used[i] -= bond[i]
                    used[i - 1] += bond[i]
                    if used[i - 1] > cap[atoms[i - 1]]:
                        smi = None
                        break
            else:
                smi += b + atoms[i]
        if smi is None:
            continue
        if ring:
            smi += '1'
        if smi not in seen:
            seen.add(smi)
            out.append(smi)
    out.sort(key=cnt_atom, reverse=True)
    return out


class Tokenizer:
    def __init__(self, vocab_path=None, num_pieces=300, pieces=None):
        """vocab_path: the reference's text format, one `smiles\\tatom_num\\tfreq` per line
        (:158-166).  pieces: the same rows as (smiles, atom_num, freq) tuples (what a checkpoint
        stores, see vocab_state).  Without either, `num_pieces` synthetic pieces are created."""
        self.vocab_dict, self.idx2piece, self.piece2idx = {}, [], {}
        self._fragments = None
        self._canonical = None
        if pieces is not None:
            for smi, atom_num, freq in pieces:
                self.vocab_dict[smi] = (int(atom_num), int(freq))
                self.piece2idx[smi] = len(self.idx2piece)
                self.idx2piece.append(smi)
        elif vocab_path is not None:
            with open(vocab_path, 'r') as fin:
                lines = fin.read().strip().split('\n')
            for line in lines:
                smi, atom_num, freq = line.strip().split('\t')
                self.vocab_dict[smi] = (int(atom_num), int(freq))
                self.piece2idx[smi] = len(self.idx2piece)
                self.idx2piece.append(smi)
        else:
            for smi in synthetic_pieces(num_pieces):
                self.vocab_dict[smi] = (cnt_atom(smi), 1)
                self.piece2idx[smi] = len(self.idx2piece)
                self.idx2piece.append(smi)
        self.pad, self.end = '<pad>', '<s>'
        for smi in [self.pad, self.end]:
            self.piece2idx[smi] = len(self.idx2piece)
            self.idx2piece.append(smi)
        self.atom_pad = '<pad>'
        self.chem_vocab = GeneralVocab(atom_special=[self.atom_pad])

    def vocab_state(self):
        """The vocabulary as plain data: [smiles, atom_num, freq] per piece, in id order (the rows
        of the vocabulary file).  Tokenizer(pieces=state) rebuilds an equal tokenizer."""
        n = len(self.idx2piece) - 2     # without <pad> and <s>
        return [[smi, self.vocab_dict[smi][0], self.vocab_dict[smi][1]] for smi in self.idx2piece[:n]]

    def canonical_pieces(self):
        """{canonical name of a piece's graph: (its SMILES in this vocabulary, its frequency)}: the
        vocabulary re-keyed by utils.molgraph.canonical_smiles, so that pieces are recognised whatever
        tool wrote the file's SMILES.  Pieces that cannot be read as kekulised SMILES are left out."""
        if self._canonical is None:
            from . import molgraph
            table = {}
            for smi in self.idx2piece[:len(self.idx2piece) - 2]:
                mol = molgraph.smiles2molecule(smi)
                if mol is not None:
                    table.setdefault(molgraph.canonical_smiles(mol.symbols, mol.bonds), (smi, self.vocab_dict[smi][1]))
            self._canonical = table
        return self._canonical

    def tokenize(self, mol, return_idx=False):
        """mol: SMILES or utils.molgraph.Mol.  [(piece SMILES, atom ids)] between two end tokens, pieces
        in shuffled order (np.random), or (piece ids, atom id groups) with return_idx."""
        from ..data import mol_bpe
        return mol_bpe.tokenize(self, mol, return_idx)

    __call__ = tokenize

    def idx_to_piece(self, idx):
        return self.idx2piece[idx]

    def piece_to_idx(self, piece):
        return self.piece2idx[piece]

    def piece_atom_counts(self):
        """cnt_atom(idx_to_piece(id)) for every piece id (src/modules/vae_piece_decoder.py:152)."""
        return [cnt_atom(smi) for smi in self.idx2piece]

    def fragment_table(self):
        """Atoms and in-piece bonds of every piece (utils.fragments.FragmentTable), parsed once."""
        if self._fragments is None:
            from .fragments import FragmentTable
            self._fragments = FragmentTable.from_tokenizer(self)
        return self._fragments

    def pad_idx(self):
        return self.piece2idx[self.pad]

    def end_idx(self):
        return self.piece2idx[self.end]

    def atom_pad_idx(self):
        return self.chem_vocab.atom_to_idx(self.atom_pad)

    def num_piece_type(self):
        return len(self.idx2piece)

    def num_atom_type(self):
        return self.chem_vocab.num_atom_type()

    def __len__(self):
        return len(self.idx2piece)
