"""Preprocessing without RDKit (utils.molgraph, data.mol_bpe): SMILES reading, kekulisation, canonical
piece names, vocabulary mining and tokenisation, on real QM9 / ZINC250K SMILES
(tests/golden/real/*.smi: the first 2048 lines of the reference's data/qm9/train.txt and
data/zinc250k/valid.txt - data files, copied as fixtures).

PARITY UNPINNED: the reference holds no vocabulary file and no tokenised molecule, and RDKit is
absent, so nothing here is compared with the reference's own output.  What is tested is the
algorithm's invariants, and the committed outputs of this implementation (the vocabulary files under
tests/golden/real/, written by `python -m ps_vae_amd.data.mol_bpe`) as a regression pin."""
import os

import numpy as np
import pytest

import ps_vae_amd  # noqa: F401
from ps_vae_amd.data import mol_bpe
from ps_vae_amd.data.packed import PackedMolecules
from ps_vae_amd.utils import molgraph as mg
from ps_vae_amd.utils.chem_vocab import MAX_VALENCE, cnt_atom
from ps_vae_amd.utils.tokenizer import Tokenizer

REAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'real')


def smiles(name, n=None):
    with open(os.path.join(REAL, name)) as f:
        out = [line.strip() for line in f if line.strip()]
    return out[:n] if n else out


@pytest.mark.parametrize('smi,atoms,bonds,doubles', [
    ('c1ccccc1', 6, 6, 3), ('c1ccncc1', 6, 6, 3), ('c1cc[nH]c1', 5, 5, 2), ('c1ccoc1', 5, 5, 2), ('c1ccsc1', 5, 5, 2),
    ('c1ccc2[nH]ccc2c1', 9, 10, 4), ('c1ccc2ccccc2c1', 10, 11, 5), ('O=c1cc[nH]cc1', 7, 7, 3),
    ('Cn1cnc2c1c(=O)n(C)c(=O)n2C', 14, 15, 4), ('C[n+]1ccccc1', 7, 7, 3), ('c1ccc2c(c1)[nH]c1ccccc12', 13, 15, 6),
    ('CC[C@H]1[C@@H](C)[C@H]1NC=O', 9, 9, 1), ('C#N', 2, 1, 0), ('ClCCBr', 4, 3, 0), ('C%12CC%12', 3, 3, 0),
])
def test_known_molecules(smi, atoms, bonds, doubles):
    mol = mg.smiles2molecule(smi)
    assert mol is not None and len(mol) == atoms and len(mol.bonds) == bonds
    assert sum(1 for _, _, o in mol.bonds if o == 2) == doubles
    assert all(o in (1, 2, 3) for _, _, o in mol.bonds) and not any(mol.aromatic)


def test_unreadable_smiles_give_none():
    for smi in ('c1cccc1', 'C1CC', 'C(C', '[H][H]', 'C[Si](C)C', 'c1ccccc1)'):
        assert mg.smiles2molecule(smi) is None, smi


@pytest.mark.parametrize('name', ['qm9_2048.smi', 'zinc_2048.smi'])
def test_every_fixture_molecule_reads_with_sane_valences(name):
    for smi in smiles(name):
        mol = mg.smiles2molecule(smi)
        assert mol is not None, smi
        adj = mol.neighbours()
        for i, sym in enumerate(mol.symbols):
            assert sum(o for _, o in adj[i]) <= MAX_VALENCE[sym] + abs(mol.charge[i]), (smi, i, sym)
        # atoms are numbered in order of appearance: as many atoms as element letters in the string
        letters = sum(1 for c in smi.replace('Cl', 'L').replace('Br', 'R') if c.isalpha() and c not in 'Hh')
        assert len(mol) == letters, smi


def test_canonical_name_does_not_depend_on_the_numbering():
    rng = np.random.default_rng(5)
    for smi in smiles('zinc_2048.smi', 60) + smiles('qm9_2048.smi', 60):
        mol = mg.smiles2molecule(smi)
        key, text = mg.canonical_key(mol.symbols, mol.bonds), mg.canonical_smiles(mol.symbols, mol.bonds)
        perm = rng.permutation(len(mol))                        # atom i becomes atom perm[i]
        symbols = [None] * len(mol)
        for i, p in enumerate(perm):
            symbols[p] = mol.symbols[i]
        bonds = [(int(perm[a]), int(perm[b]), o) for a, b, o in mol.bonds]
        rng.shuffle(bonds)
        assert mg.canonical_key(symbols, bonds) == key, smi
        assert mg.canonical_smiles(symbols, bonds) == text, smi
        again = mg.smiles2molecule(text)                        # the name is SMILES of the same graph, and a fixed point
        assert again is not None and mg.canonical_key(again.symbols, again.bonds) == key, (smi, text)
        assert mg.canonical_smiles(again.symbols, again.bonds) == text, (smi, text)
        assert cnt_atom(text) == len(mol), (smi, text)


def test_different_graphs_get_different_names():
    names = {}
    for smi in ('CCO', 'COC', 'CC=O', 'C=CO', 'C1CC1', 'CCC', 'C1=CC=CC=C1', 'C1CCCCC1', 'C1=CCCCC1', 'CC(C)C', 'CCCC', 'C1CCC1'):
        mol = mg.smiles2molecule(smi)
        names.setdefault(mg.canonical_smiles(mol.symbols, mol.bonds), []).append(smi)
    assert all(len(v) == 1 for v in names.values()), names


def test_vocabulary_mining_reproduces_the_committed_file(tmp_path):
    out = tmp_path / 'vocab.txt'
    pieces, details = mol_bpe.graph_bpe(None, 60, str(out), smiles=smiles('zinc_2048.smi', 512))
    with open(os.path.join(REAL, 'zinc_vocab_60_of_512.txt')) as f:
        assert out.read_text() == f.read()
    assert len(pieces) == 60 and set(MAX_VALENCE) <= set(pieces)
    counts = [details[p][0] for p in pieces]
    assert counts == sorted(counts, reverse=True)              # largest pieces first (mol_bpe.py:148-149)
    assert all(details[p][0] == cnt_atom(p) for p in pieces)


@pytest.mark.parametrize('corpus,vocab', [('zinc_2048.smi', 'zinc_vocab_300.txt'), ('qm9_2048.smi', 'qm9_vocab_300.txt')])
def test_tokenisation_invariants(corpus, vocab):
    tok = Tokenizer(os.path.join(REAL, vocab))
    assert tok.num_piece_type() == 302
    known = tok.canonical_pieces()
    # every piece of the file reads back under its own name (a piece can be mined twice, as in the reference:
    # a merge skipped because its partner was just absorbed comes up again in a later round)
    assert len(known) == len(set(tok.idx2piece[:300])) >= 298 and all(k == v[0] for k, v in known.items())
    used = set()
    for smi in smiles(corpus, 150):
        mol = mg.smiles2molecule(smi)
        np.random.seed(3)
        res = tok.tokenize(smi)
        np.random.seed(3)
        ids, groups = tok.tokenize(mol, return_idx=True)
        assert res[0] == (tok.end, []) and res[-1] == (tok.end, [])
        assert [tok.piece_to_idx(p) for p, _ in res] == ids and [g for _, g in res] == groups
        assert sorted(a for g in groups for a in g) == list(range(len(mol))), smi      # the pieces partition the atoms
        adj = [[j for j, _ in row] for row in mol.neighbours()]
        for piece, group in res[1:-1]:
            assert piece in tok.vocab_dict
            seen, todo = {group[0]}, [group[0]]                 # every piece is connected
            while todo:
                for j in adj[todo.pop()]:
                    if j in group and j not in seen:
                        seen.add(j)
                        todo.append(j)
            assert seen == set(group), (smi, piece, group)
        used.update(ids)
    assert len(used) > 40                                       # the mined pieces do occur


def test_process_step1_feeds_the_packed_layout_and_the_oracle_collation():
    from oracle import collate_ref
    tok = Tokenizer(os.path.join(REAL, 'zinc_vocab_300.txt'))
    np.random.seed(0)
    data = [mol_bpe.process_step1(smi, tok) for smi in smiles('zinc_2048.smi', 16)]
    for d, smi in zip(data, smiles('zinc_2048.smi', 16)):
        mol = mg.smiles2molecule(smi)
        assert len(d['x']) == len(mol) and len(d['edge_index']) == len(mol.bonds) == len(d['edge_attr'])
        assert d['pieces'][0] == d['pieces'][-1] == tok.end_idx() and d['groups'][0] == d['groups'][-1] == []
        assert set(d['edge_attr']) <= {1, 2, 3}
    packed = PackedMolecules.from_step1(data)
    assert len(packed) == 16
    batch, used = collate_ref.collate(data, np.random.default_rng(1).random(20000), piece_pad=tok.pad_idx())
    assert batch['x'].shape[0] == 16 and used > 0
