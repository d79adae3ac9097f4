"""Vocabulary-file Tokenizer surface needed by the hot path (src/data/mol_bpe.py:156-249):
special-token layout and id lookups.  Tokenising a molecule (the greedy principal-subgraph
merge, :175-222) needs RDKit canonical SMILES and is outside the path."""
from .chem_vocab import GeneralVocab, cnt_atom


class Tokenizer:
    def __init__(self, vocab_path=None, num_pieces=300):
        """vocab_path: the reference's text format, one `smiles\\tatom_num\\tfreq` per line
        (:158-166).  Without a file, `num_pieces` anonymous pieces are created."""
        self.vocab_dict, self.idx2piece, self.piece2idx = {}, [], {}
        if vocab_path is not None:
            with open(vocab_path, 'r') as fin:
                lines = fin.read().strip().split('\n')
            for line in lines:
                smi, atom_num, freq = line.strip().split('\t')
                self.vocab_dict[smi] = (int(atom_num), int(freq))
                self.piece2idx[smi] = len(self.idx2piece)
                self.idx2piece.append(smi)
        else:
            for i in range(num_pieces):
                self.piece2idx['P%d' % i] = i
                self.idx2piece.append('P%d' % i)
        self.pad, self.end = '<pad>', '<s>'
        for smi in [self.pad, self.end]:
            self.piece2idx[smi] = len(self.idx2piece)
            self.idx2piece.append(smi)
        self.atom_pad = '<pad>'
        self.chem_vocab = GeneralVocab(atom_special=[self.atom_pad])

    def tokenize(self, mol, return_idx=False):
        raise NotImplementedError('principal-subgraph tokenisation needs RDKit (src/data/mol_bpe.py:175-222)')

    __call__ = tokenize

    def idx_to_piece(self, idx):
        return self.idx2piece[idx]

    def piece_to_idx(self, piece):
        return self.piece2idx[piece]

    def piece_atom_counts(self):
        """cnt_atom(idx_to_piece(id)) for every piece id (src/modules/vae_piece_decoder.py:152)."""
        return [cnt_atom(smi) for smi in self.idx2piece]

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
