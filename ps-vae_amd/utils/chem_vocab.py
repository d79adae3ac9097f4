"""Atom / bond index tables of the hot path (reference: src/utils/chem_utils.py:14-63).
Only the index meaning is needed here; the RDKit conversions of the reference's
chem_utils are outside the path."""

ATOMS = ['B', 'Br', 'C', 'Cl', 'F', 'I', 'N', 'O', 'P', 'S']   # key order of MAX_VALENCE, :14
MAX_VALENCE = {'B': 3, 'Br': 1, 'C': 4, 'Cl': 1, 'F': 1, 'I': 1, 'N': 3, 'O': 2, 'P': 5, 'S': 6}
BONDS = [None, 'SINGLE', 'DOUBLE', 'TRIPLE']                    # Bond_List, :15


class GeneralVocab:
    """Same methods as the reference's GeneralVocab (:18-63); bonds are named by string
    ('SINGLE' ...) instead of rdkit BondType members, None stays None (index 0)."""

    def __init__(self, atom_special=None, bond_special=None):
        self.atom_special = list(atom_special or [])
        self.bond_special = list(bond_special or [])
        self.idx2atom = ATOMS + self.atom_special
        self.atom2idx = {a: i for i, a in enumerate(self.idx2atom)}
        self.idx2bond = BONDS + self.bond_special
        self.bond2idx = {b: i for i, b in enumerate(self.idx2bond)}

    def idx_to_atom(self, idx):
        return self.idx2atom[idx]

    def atom_to_idx(self, atom):
        return self.atom2idx[atom]

    def idx_to_bond(self, idx):
        return self.idx2bond[idx]

    def bond_to_idx(self, bond):
        name = getattr(bond, 'name', bond)      # accepts rdkit BondType members too
        return self.bond2idx[name]

    def bond_idx_to_valence(self, idx):
        return {'SINGLE': 1, 'DOUBLE': 2, 'TRIPLE': 3}.get(self.idx2bond[idx], -1)

    def num_atom_type(self):
        return len(self.idx2atom)

    def num_bond_type(self):
        return len(self.idx2bond)


def cnt_atom(smi):
    """Characters of a (kekulised) piece SMILES that are keys of MAX_VALENCE
    (src/utils/chem_utils.py:158-163): the atom count the sampling loop budgets with."""
    return sum(1 for c in smi if c in MAX_VALENCE)
