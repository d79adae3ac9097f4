import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # Multi-rank GPU tests start their ranks from a fork server.  It has to exist before any test
    # initialises the GPU in this process: from then on this process must not exec another program.
    import multiprocessing
    from multiprocessing import forkserver
    multiprocessing.set_forkserver_preload([])
    forkserver.ensure_running()


class Tokenizer:
    """Tokenizer stand-in with the reference's special-token layout
    (src/data/mol_bpe.py:167-173): pieces, then '<pad>', then '<s>'; 10 atoms + '<pad>'."""

    def __init__(self, num_pieces=300):
        from ps_vae_amd.utils.chem_vocab import GeneralVocab
        self.num_pieces = num_pieces
        self.chem_vocab = GeneralVocab(atom_special=['<pad>'])

    def pad_idx(self):
        return self.num_pieces

    def end_idx(self):
        return self.num_pieces + 1

    def atom_pad_idx(self):
        return self.chem_vocab.atom_to_idx('<pad>')

    def num_piece_type(self):
        return self.num_pieces + 2

    def num_atom_type(self):
        return self.chem_vocab.num_atom_type()

    def idx_to_piece(self, idx):
        return 'P%d' % idx if idx < self.num_pieces else ('<pad>', '<s>')[idx - self.num_pieces]


@pytest.fixture(scope='session')
def tokenizer():
    return Tokenizer()


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def sub(d, prefix):
    """Entries of d under 'prefix/' with the prefix stripped."""
    p = prefix + '/'
    return {k[len(p):]: v for k, v in d.items() if k.startswith(p)}
