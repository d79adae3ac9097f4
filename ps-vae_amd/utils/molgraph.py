"""Molecules as plain graphs, without RDKit: SMILES in, kekulised atoms + bonds out, and a canonical
key / SMILES for sub-graphs.

The reference reaches all of this through RDKit (src/utils/chem_utils.py:66-78, 132-156:
Chem.MolFromSmiles, RemoveStereochemistry, Kekulize, MolToSmiles of a sub-molecule rebuilt from
element symbols and bond types).  What the data path needs from it is small:

  * atoms numbered in order of appearance in the SMILES, element symbol per atom
    (src/data/bpe_dataset.py:34-35), bonds with their kekulised type 1..3 (:38-43);
  * "these two sub-graphs are the same piece": the reference compares canonical SMILES of
    sub-molecules that carry ONLY element symbols and bond types (get_submol drops charges,
    hydrogens and aromaticity), i.e. it compares isomorphism classes of (symbol, bond order)
    labelled graphs.  `canonical_key` returns one string per such class and `canonical_smiles`
    a kekulised SMILES of it, so a vocabulary file written by the reference can be read (its
    SMILES are parsed and keyed the same way) and one written here can be read by the reference.

Not reproducible without RDKit, and said so where it matters: WHICH Kekule structure of an aromatic
system is taken (any valid one is produced here, by a deterministic search in atom order; RDKit's
choice can differ, and with it the pieces inside aromatic rings), the order of an atom's bonds
(ring-closure bonds are appended where the ring closes), and the text of a canonical SMILES.
"""
from .chem_vocab import MAX_VALENCE

_BOND_SYMBOL = {'-': 1, '=': 2, '#': 3, '/': 1, '\\': 1, ':': 4}   # 4: aromatic, resolved by kekulise()
_AROMATIC = {'b': 'B', 'c': 'C', 'n': 'N', 'o': 'O', 'p': 'P', 's': 'S'}
_ORGANIC = ('Cl', 'Br', 'B', 'C', 'N', 'O', 'P', 'S', 'F', 'I')
# valences an element may take (lowest first), neutral; a positive charge on N / P / O / S adds one,
# a negative charge takes one away (and the reverse for B / C)
_VALENCES = {'B': (3,), 'C': (4,), 'N': (3, 5), 'O': (2,), 'P': (3, 5), 'S': (2, 4, 6), 'F': (1,), 'Cl': (1,), 'Br': (1,),
             'I': (1,)}


class Mol:
    """symbols[i], aromatic[i], charge[i], hcount[i] (explicit H count of a bracket atom, None outside
    brackets); bonds [(a, b, order)] in the order they were read, order 4 = aromatic until kekulise()."""

    def __init__(self, smiles=''):
        self.smiles = smiles
        self.symbols, self.aromatic, self.charge, self.hcount = [], [], [], []
        self.bonds = []

    def __len__(self):
        return len(self.symbols)

    def neighbours(self):
        """adj[i] = [(j, order)] in bond order."""
        adj = [[] for _ in self.symbols]
        for a, b, o in self.bonds:
            adj[a].append((b, o))
            adj[b].append((a, o))
        return adj


def parse_smiles(smi):
    """SMILES -> Mol with atoms in order of appearance.  Stereo marks, isotopes and atom classes are
    dropped (Chem.RemoveStereochemistry at chem_utils.py:71); explicit hydrogen ATOMS are refused
    (the datasets of the reference have none)."""
    mol = Mol(smi)
    stack, prev, pending = [], None, None
    rings = {}
    i, n = 0, len(smi)

    def add_atom(sym, aromatic, charge, hcount):
        nonlocal prev, pending
        if sym not in MAX_VALENCE:
            raise ValueError(f'{smi!r}: element {sym!r} is not in the chemical vocabulary')
        mol.symbols.append(sym)
        mol.aromatic.append(aromatic)
        mol.charge.append(charge)
        mol.hcount.append(hcount)
        cur = len(mol.symbols) - 1
        if prev is not None:
            order = pending or (4 if aromatic and mol.aromatic[prev] else 1)
            mol.bonds.append((prev, cur, order))
        prev, pending = cur, None

    def ring(label):
        nonlocal pending
        if prev is None:
            raise ValueError(f'{smi!r}: ring closure before any atom')
        if label in rings:
            other, order = rings.pop(label)
            order = pending or order or (4 if mol.aromatic[other] and mol.aromatic[prev] else 1)
            mol.bonds.append((other, prev, order))
        else:
            rings[label] = (prev, pending)
        pending = None

    while i < n:
        c = smi[i]
        if c == '[':
            j = smi.find(']', i)
            if j < 0:
                raise ValueError(f'{smi!r}: unclosed bracket atom')
            body = smi[i + 1:j].lstrip('0123456789')
            if body[:1].isupper():
                width = 2 if body[1:2].islower() else 1          # Cl, Br, Si, Se, ... (unknown elements are refused below)
                sym, aromatic, rest = body[:width], False, body[width:]
                if sym == 'H':
                    raise ValueError(f'{smi!r}: explicit hydrogen atom')
            elif body[:1] in _AROMATIC and not body[1:2].islower():
                sym, aromatic, rest = _AROMATIC[body[0]], True, body[1:]
            else:
                raise ValueError(f'{smi!r}: cannot read the bracket atom [{body}]')
            rest = rest.lstrip('@')
            if rest[:2] in ('TH', 'AL', 'SP', 'TB', 'OH'):
                rest = rest[2:].lstrip('0123456789')
            hcount = 0
            if rest[:1] == 'H':
                rest = rest[1:]
                digits = ''
                while rest[:1].isdigit():
                    digits, rest = digits + rest[0], rest[1:]
                hcount = int(digits) if digits else 1
            charge = 0
            while rest[:1] in ('+', '-'):
                sign = 1 if rest[0] == '+' else -1
                rest = rest[1:]
                digits = ''
                while rest[:1].isdigit():
                    digits, rest = digits + rest[0], rest[1:]
                charge += sign * (int(digits) if digits else 1)
            add_atom(sym, aromatic, charge, hcount)
            i = j + 1
        elif smi[i:i + 2] in ('Cl', 'Br'):
            add_atom(smi[i:i + 2], False, 0, None)
            i += 2
        elif c in _AROMATIC:
            add_atom(_AROMATIC[c], True, 0, None)
            i += 1
        elif c.isalpha():
            add_atom(c, False, 0, None)
            i += 1
        elif c in _BOND_SYMBOL:
            pending = _BOND_SYMBOL[c]
            i += 1
        elif c == '(':
            stack.append(prev)
            i += 1
        elif c == ')':
            if not stack:
                raise ValueError(f'{smi!r}: unbalanced parenthesis')
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
            raise ValueError(f'{smi!r}: unexpected character {c!r}')
    if rings or stack:
        raise ValueError(f'{smi!r}: unclosed ring or branch')
    return mol


def _wants_double(mol, adj, i):
    """Does aromatic atom i take one of the ring's double bonds?  It does when, with every aromatic bond
    counted as single, one unit of its lowest valence is left over (pyridine n: 3 - 2; pyrrole [nH]: 3 - 3,
    no; furan o, thiophene s: 2 - 2, no; a carbon with an exocyclic double bond: 4 - 4, no).  A charge moves
    the valence (N+, O+: one more; C-, N-: one less).  Outside brackets the hydrogens are implicit: an
    aromatic carbon with two or three ring bonds keeps one unit for the double bond either way."""
    sym, q = mol.symbols[i], mol.charge[i]
    used = sum(1 if o == 4 else o for _, o in adj[i])
    valence = _VALENCES[sym][0] + (q if sym in ('N', 'P', 'O', 'S') else -abs(q) if sym in ('B', 'C') else 0)
    if mol.hcount[i] is not None:                        # bracket atom: hydrogens are as written
        return valence - used - mol.hcount[i] == 1
    return valence - used >= 1 if sym in ('C', 'B') else valence - used == 1


def kekulise(mol):
    """Replace the aromatic bonds by single / double bonds (Chem.Kekulize at chem_utils.py:73): a
    perfect matching of the atoms that take a double bond, found by a deterministic search in atom
    order (an atom's candidates in the order of its bonds).  Raises ValueError when there is none."""
    if not any(o == 4 for _, _, o in mol.bonds):
        mol.aromatic = [False] * len(mol)
        return mol
    adj = mol.neighbours()
    wants = [mol.aromatic[i] and _wants_double(mol, adj, i) for i in range(len(mol))]
    cand = [[j for j, o in adj[i] if o == 4 and wants[j]] if wants[i] else [] for i in range(len(mol))]
    mate = [-1] * len(mol)
    todo = [i for i in range(len(mol)) if wants[i]]

    def search(pos):
        while pos < len(todo) and mate[todo[pos]] >= 0:
            pos += 1
        if pos == len(todo):
            return True
        i = todo[pos]
        for j in cand[i]:
            if mate[j] < 0:
                mate[i], mate[j] = j, i
                if search(pos + 1):
                    return True
                mate[i], mate[j] = -1, -1
        return False

    if not search(0):
        raise ValueError(f'{mol.smiles!r}: no Kekule structure')
    mol.bonds = [(a, b, (2 if mate[a] == b else 1) if o == 4 else o) for a, b, o in mol.bonds]
    mol.aromatic = [False] * len(mol)
    return mol


def smiles2molecule(smiles, kekulize=True):
    """The reference's helper of the same name (chem_utils.py:66-74) on plain graphs; None when the
    string cannot be read or kekulised (RDKit returns None / raises there)."""
    try:
        mol = parse_smiles(smiles)
        return kekulise(mol) if kekulize else mol
    except ValueError:
        return None


# ------------------------------------------------------------------ canonical form of a labelled sub-graph
def _refine(colours, adj):
    """Colour refinement to a stable partition: a node's new colour = (old colour, sorted multiset of
    (bond order, neighbour colour)); colours are renumbered by sorted signature, so they do not depend
    on the numbering of the nodes."""
    n = len(colours)
    while True:
        sigs = [(colours[i], tuple(sorted((o, colours[j]) for j, o in adj[i]))) for i in range(n)]
        rank = {s: r for r, s in enumerate(sorted(set(sigs)))}
        new = [rank[s] for s in sigs]
        if len(rank) == len(set(colours)):
            return new
        colours = new


def _canonical_order(labels, adj):
    """The numbering of the nodes (a permutation `order`, order[k] = node placed k-th) whose
    (labels, bond list) string is smallest among those individualisation-refinement reaches."""
    n = len(labels)
    rank = {s: r for r, s in enumerate(sorted(set(labels)))}
    start = _refine([rank[s] for s in labels], adj)
    best = [None, None]

    def code(colours):
        order = sorted(range(n), key=lambda i: colours[i])
        pos = {v: k for k, v in enumerate(order)}
        edges = sorted((min(pos[i], pos[j]), max(pos[i], pos[j]), o) for i in range(n) for j, o in adj[i] if i < j)
        return (tuple(labels[i] for i in order), tuple(edges)), order

    def descend(colours):
        cells = {}
        for i, c in enumerate(colours):
            cells.setdefault(c, []).append(i)
        split = [c for c in sorted(cells) if len(cells[c]) > 1]
        if not split:
            key, order = code(colours)
            if best[0] is None or key < best[0]:
                best[0], best[1] = key, order
            return
        c = split[0]
        for v in cells[c]:
            # individualise v: it sorts before the rest of its cell
            trial = [2 * x + (1 if (x == c and i != v) else 0) for i, x in enumerate(colours)]
            rank2 = {s: r for r, s in enumerate(sorted(set(trial)))}
            descend(_refine([rank2[x] for x in trial], adj))

    descend(start)
    return best[0], best[1]


def _sub_graph(symbols, bonds, atoms):
    atoms = list(atoms)
    local = {a: k for k, a in enumerate(atoms)}
    adj = [[] for _ in atoms]
    for a, b, o in bonds:
        if a in local and b in local:
            adj[local[a]].append((local[b], o))
            adj[local[b]].append((local[a], o))
    return [symbols[a] for a in atoms], adj


def canonical_key(symbols, bonds, atoms=None):
    """One hashable key per isomorphism class of the sub-graph induced by `atoms` (default: all), with
    element symbols on the nodes and bond orders on the edges - what equality of the reference's
    canonical sub-molecule SMILES decides (mol_bpe.py:63-64)."""
    labels, adj = _sub_graph(symbols, bonds, range(len(symbols)) if atoms is None else atoms)
    if len(labels) == 1:
        return ((labels[0],), ())
    key, _ = _canonical_order(labels, adj)
    return key


_ORDER_CHAR = {1: '', 2: '=', 3: '#'}


def canonical_smiles(symbols, bonds, atoms=None):
    """A kekulised SMILES of the sub-graph, the same string for every member of its isomorphism class
    (depth-first from the first atom of the canonical numbering, neighbours in canonical order).
    It is valid SMILES for any reader; it is not RDKit's canonical text."""
    labels, adj = _sub_graph(symbols, bonds, range(len(symbols)) if atoms is None else atoms)
    n = len(labels)
    if n == 1:
        return _atom_text(labels[0])
    _, order = _canonical_order(labels, adj)
    pos = {v: k for k, v in enumerate(order)}
    nbrs = [sorted(adj[i], key=lambda t: pos[t[0]]) for i in range(n)]
    # first pass: spanning tree and ring-closure bonds
    seen, parent, closures = set(), {}, {}
    stack = [(order[0], -1)]
    visit = []
    while stack:
        v, p = stack.pop()
        if v in seen:
            continue
        seen.add(v)
        parent[v] = p
        visit.append(v)
        for u, _ in reversed(nbrs[v]):
            if u not in seen:
                stack.append((u, v))
    tree = {(parent[v], v) for v in visit if parent[v] >= 0}
    labels_of = [[] for _ in range(n)]
    next_label = 1
    when = {v: k for k, v in enumerate(visit)}
    for v in visit:
        for u, o in nbrs[v]:
            if (v, u) in tree or (u, v) in tree or when[u] < when[v]:
                continue
            labels_of[v].append((next_label, o, True))
            labels_of[u].append((next_label, o, False))
            next_label += 1

    def ring_text(v):
        out = ''
        for lab, o, opening in labels_of[v]:
            out += (_ORDER_CHAR[o] if opening else '') + (str(lab) if lab < 10 else f'%{lab:02d}')
        return out

    def write(v):
        text = _atom_text(labels[v]) + ring_text(v)
        kids = [(u, o) for u, o in nbrs[v] if parent.get(u) == v and (v, u) in tree]
        for k, (u, o) in enumerate(kids):
            part = _ORDER_CHAR[o] + write(u)
            text += part if k == len(kids) - 1 else '(' + part + ')'
        return text

    return write(order[0])


def _atom_text(sym):
    return sym if sym in _ORGANIC else f'[{sym}]'


def get_submol_key(mol, atoms):
    return canonical_key(mol.symbols, mol.bonds, atoms)
