"""Principal-subgraph vocabulary mining and tokenisation without RDKit
(reference: src/data/mol_bpe.py:20-153 MolInPiece / freq_cnt / graph_bpe, :175-222 Tokenizer.tokenize;
src/data/bpe_dataset.py:30-54 process_step1).

Same algorithm, same data structures and iteration orders as the reference; what RDKit did there -
reading the SMILES, kekulising, naming a sub-graph by a canonical SMILES - is utils.molgraph here.
A piece is identified by the isomorphism class of its (element, bond order) labelled graph, which
is what equality of the reference's sub-molecule SMILES decides, so a vocabulary file mined by the
reference tokenises the same here (its SMILES are re-keyed when read) up to the choice of Kekule
structure inside aromatic systems, which only RDKit can reproduce; no output of the reference's
tokenizer ships with it, so this module's results are not pinned by a reference fixture.
"""
from copy import copy

import numpy as np

from ..utils import molgraph
from ..utils.chem_vocab import MAX_VALENCE, cnt_atom


class MolInPiece:
    """A molecule as a set of pieces that grows by merging neighbouring pieces (mol_bpe.py:20-96)."""

    def __init__(self, mol):
        self.mol = mol
        self.adj = [[j for j, _ in row] for row in mol.neighbours()]
        self.pieces, self.pieces_smis = {}, {}      # piece id (its first atom's id) -> {atom id: symbol} / SMILES
        for idx, symbol in enumerate(mol.symbols):
            self.pieces[idx] = {idx: symbol}
            self.pieces_smis[idx] = symbol
        self.inversed_index = {aid: aid for aid in range(len(mol))}
        self.dirty = True
        self.smi2pids = {}
        self._names = {}                            # frozenset of atom ids -> canonical SMILES (never changes)

    def get_nei_pieces(self):
        nei_pieces, merge_pids = [], []
        for key in self.pieces:
            piece = self.pieces[key]
            local_nei_pid = []
            for aid in piece:
                for nei_idx in self.adj[aid]:
                    if nei_idx in piece or nei_idx > aid:     # only consider connecting to former atoms
                        continue
                    local_nei_pid.append(self.inversed_index[nei_idx])
            for nei_pid in set(local_nei_pid):
                new_piece = copy(piece)
                new_piece.update(self.pieces[nei_pid])
                nei_pieces.append(new_piece)
                merge_pids.append((key, nei_pid))
        return nei_pieces, merge_pids

    def _name(self, piece):
        atoms = frozenset(piece)
        name = self._names.get(atoms)
        if name is None:
            name = molgraph.canonical_smiles(self.mol.symbols, self.mol.bonds, piece.keys())
            self._names[atoms] = name
        return name

    def get_nei_smis(self):
        if self.dirty:
            nei_pieces, merge_pids = self.get_nei_pieces()
            self.smi2pids = {}
            for piece, pids in zip(nei_pieces, merge_pids):
                self.smi2pids.setdefault(self._name(piece), []).append(pids)
            self.dirty = False
        return list(self.smi2pids.keys())

    def merge(self, smi):
        if self.dirty:
            self.get_nei_smis()
        if smi in self.smi2pids:
            for pid1, pid2 in self.smi2pids[smi]:
                if pid1 in self.pieces and pid2 in self.pieces:   # possibly deleted by a former merge
                    self.pieces[pid1].update(self.pieces[pid2])
                    self.pieces_smis[pid1] = smi
                    for aid in self.pieces[pid2]:
                        self.inversed_index[aid] = pid1
                    del self.pieces[pid2]
                    del self.pieces_smis[pid2]
        self.dirty = True

    def get_smis_pieces(self):
        return [(self.pieces_smis[pid], list(self.pieces[pid].keys())) for pid in self.pieces_smis]


def freq_cnt(mol):
    freqs = {}
    for smi in mol.get_nei_smis():
        freqs[smi] = freqs.get(smi, 0) + 1
    return freqs, mol


def graph_bpe(fname, vocab_len, vocab_path=None, cpus=1, smiles=None, log=None):
    """Mine `vocab_len` pieces from the SMILES file `fname` (or the list `smiles`): the 10 atoms, then
    the most frequent neighbouring-piece merge, one per round (mol_bpe.py:108-153).  Returns
    (pieces sorted by atom count, details {piece: [atom count, frequency]}) and writes the
    `smiles\\tatoms\\tfrequency` text file when vocab_path is given.  Molecules with a '+' in their
    SMILES are left out, as the reference leaves them out; unreadable ones are skipped.  `cpus` is the
    reference's worker count: accepted and not used, the rounds run in this process."""
    if smiles is None:
        with open(fname, 'r') as fin:
            smiles = [line.strip() for line in fin if line.strip()]
    mols = []
    for smi in smiles:
        if '+' in smi:
            continue
        mol = molgraph.smiles2molecule(smi)
        if mol is not None:
            mols.append(MolInPiece(mol))
    selected_smis, details = list(MAX_VALENCE.keys()), {}
    for atom in selected_smis:
        details[atom] = [1, 0]
    for smi in smiles:                 # (single characters: 'C' also counts the C of Cl, as in the reference)
        for c in smi:
            if c in details:
                details[c][1] += 1
    for it in range(vocab_len - len(selected_smis)):
        freqs = {}
        for mol in mols:
            for key, cnt in freq_cnt(mol)[0].items():
                freqs[key] = freqs.get(key, 0) + cnt
        max_cnt, merge_smi = 0, ''
        for smi, cnt in freqs.items():
            if cnt > max_cnt:
                max_cnt, merge_smi = cnt, smi
        if max_cnt == 0:
            break                      # nothing left to merge: the corpus is smaller than the vocabulary asked for
        for mol in mols:
            mol.merge(merge_smi)
        selected_smis.append(merge_smi)
        details[merge_smi] = [cnt_atom(merge_smi), max_cnt]
        if log is not None and it % 20 == 0:
            log(f'round {it}: {merge_smi} x {max_cnt}')
    selected_smis.sort(key=lambda x: details[x][0], reverse=True)
    if vocab_path is not None:
        with open(vocab_path, 'w') as fout:
            fout.writelines(f'{smi}\t{details[smi][0]}\t{details[smi][1]}\n' for smi in selected_smis)
    return selected_smis, details


def tokenize(tokenizer, mol, return_idx=False):
    """Tokenizer.tokenize (mol_bpe.py:175-222): greedy merging by vocabulary frequency, then the
    pieces in shuffled order (np.random.shuffle, as the reference) between two end tokens.
    mol: SMILES or molgraph.Mol (kekulised)."""
    if isinstance(mol, str):
        parsed = molgraph.smiles2molecule(mol)
        if parsed is None:
            raise ValueError(f'cannot read or kekulise {mol!r}')
        mol = parsed
    pieces = MolInPiece(mol)
    known = tokenizer.canonical_pieces()        # canonical name -> (vocabulary SMILES, frequency)
    while True:
        max_freq, merge_smi = -1, ''
        for smi in pieces.get_nei_smis():
            hit = known.get(smi)
            if hit is not None and hit[1] > max_freq:
                max_freq, merge_smi = hit[1], smi
        if max_freq == -1:
            break
        pieces.merge(merge_smi)
    res = pieces.get_smis_pieces()
    np.random.shuffle(res)
    res = [(known[smi][0] if smi in known else smi, aids) for smi, aids in res]
    res.insert(0, (tokenizer.end, []))
    res.append((tokenizer.end, []))
    if not return_idx:
        return res
    return [tokenizer.piece_to_idx(x[0]) for x in res], [x[1] for x in res]


BOND_NAME = {1: 'SINGLE', 2: 'DOUBLE', 3: 'TRIPLE'}


def process_step1(mol, tokenizer, props=None):
    """BPEMolDataset.process_step1 (bpe_dataset.py:30-54) on a molgraph.Mol or SMILES.  The three
    property scores (QED, SA, logP: evaluation.utils.get_normalized_property_scores) are RDKit
    descriptors; they are taken from `props` and default to zeros."""
    if isinstance(mol, str):
        parsed = molgraph.smiles2molecule(mol)
        if parsed is None:
            raise ValueError(f'cannot read or kekulise {mol!r}')
        mol = parsed
    vocab = tokenizer.chem_vocab
    pieces, groups = tokenize(tokenizer, mol, return_idx=True)
    return {
        'x': [vocab.atom_to_idx(s) for s in mol.symbols],
        'edge_index': [[a, b] for a, b, _ in mol.bonds],
        'edge_attr': [vocab.bond_to_idx(BOND_NAME[o]) for _, _, o in mol.bonds],
        'props': [0.0, 0.0, 0.0] if props is None else [float(v) for v in props],
        'pieces': pieces,
        'groups': groups,
    }


def parse(argv=None):
    """The reference's command line (mol_bpe.py:252-260)."""
    import argparse
    parser = argparse.ArgumentParser(description='Graph bpe')
    parser.add_argument('--smiles', type=str, default='O=C(Cc1cc(C(F)(F)F)cc(C(F)(F)F)c1)NCC(c1ccccc1Br)N1CCC(N2CCCCC2)CC1',
                        help='The molecule to tokenize (example)')
    parser.add_argument('--data', type=str, required=True, help='Path to molecule corpus')
    parser.add_argument('--vocab_size', type=int, default=500, help='Length of vocab')
    parser.add_argument('--output', type=str, required=True, help='Path to save vocab')
    parser.add_argument('--workers', type=int, default=1, help='(kept for the reference\'s command line; mining runs in one process)')
    return parser.parse_args(argv)


def main(argv=None):
    from ..utils.tokenizer import Tokenizer
    args = parse(argv)
    graph_bpe(args.data, vocab_len=args.vocab_size, vocab_path=args.output, cpus=args.workers, log=print)
    tokenizer = Tokenizer(args.output)
    print(f'Example: {args.smiles}')
    print(tokenizer.tokenize(args.smiles))


if __name__ == '__main__':
    main()
