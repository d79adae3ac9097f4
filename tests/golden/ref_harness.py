"""Import the reference's own Python files (read-only, from /root/reference/src) on a
machine that lacks torch_geometric / rdkit / pytorch_lightning.

This file is used ONLY by tests/golden/make_golden.py, in the build container, to
produce the fixture files committed next to it.  Nothing here (and nothing under
/root/reference) is needed or read on the GPU box.

The three missing third-party packages are replaced by minimal stand-ins registered
in sys.modules *before* the reference modules are imported.  The only arithmetic
restated here is torch_geometric.nn.GINEConv (third-party, un-vendored, version
unpinned by the reference): out = nn((1 + eps) * x_i + sum_{j->i} ReLU(x_j + e_ji)),
eps = 0 (non-trainable buffer), aggregation "add", flow source_to_target
(x_j = x[edge_index[0]], summed into edge_index[1]).  Call sites in the reference:
src/modules/encoder.py:48 (constructor) and :58 (forward).
"""
import enum
import os
import sys
import types

import torch
import torch.nn as nn

REF_SRC = os.environ.get('PSVAE_REFERENCE_SRC', '/root/reference/src')


class _GINEConv(nn.Module):
    def __init__(self, nn_module, eps=0.0, train_eps=False):
        super().__init__()
        self.nn = nn_module
        self.register_buffer('eps', torch.tensor([eps]))

    def forward(self, x, edge_index, edge_attr):
        msg = torch.relu(x[edge_index[0]] + edge_attr)
        agg = torch.zeros_like(x).index_add_(0, edge_index[1], msg)
        return self.nn(agg + (1 + self.eps) * x)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    if 'torch_geometric' in sys.modules:
        return
    # torch_geometric
    _mod('torch_geometric')
    _mod('torch_geometric.nn', GINEConv=_GINEConv)
    _mod('torch_geometric.data', Data=object, Batch=object)

    # rdkit (only names looked up at import time)
    class BondType(enum.Enum):
        SINGLE = 1
        DOUBLE = 2
        TRIPLE = 3
        AROMATIC = 12

    rdchem = _mod('rdkit.Chem.rdchem', BondType=BondType)
    draw = _mod('rdkit.Chem.Draw')
    qed = _mod('rdkit.Chem.QED', qed=lambda mol: 0.0)
    allchem = _mod('rdkit.Chem.AllChem')
    desc = _mod('rdkit.Chem.Descriptors')
    rdconfig = _mod('rdkit.Chem.RDConfig', RDContribDir='/nonexistent')
    chem = _mod('rdkit.Chem', rdchem=rdchem, Draw=draw, QED=qed, AllChem=allchem,
                Descriptors=desc, RDConfig=rdconfig, BondType=BondType)
    datastructs = _mod('rdkit.DataStructs')
    _mod('rdkit', Chem=chem, DataStructs=datastructs)
    _mod('sascorer')

    # pytorch_lightning
    class _LightningModule(nn.Module):
        global_step = 0

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    es = _mod('pytorch_lightning.callbacks.early_stopping', EarlyStopping=object)
    cbs = _mod('pytorch_lightning.callbacks', early_stopping=es, EarlyStopping=object,
               ModelCheckpoint=object)
    _mod('pytorch_lightning', LightningModule=_LightningModule, callbacks=cbs,
         Trainer=object)


def import_reference():
    """Returns a namespace holding the reference's classes, imported unmodified."""
    install_stubs()
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    from modules.common_nn import MLP
    from modules.encoder import Encoder
    from modules.predictor import Predictor
    from modules.vae_piece_decoder import VAEPieceDecoder
    from data.bpe_dataset import BPEMolDataset
    from utils.chem_utils import GeneralVocab
    from pl_models.ps_vae_model import PSVAEModel
    ns = types.SimpleNamespace(MLP=MLP, Encoder=Encoder, Predictor=Predictor,
                               VAEPieceDecoder=VAEPieceDecoder, BPEMolDataset=BPEMolDataset,
                               GeneralVocab=GeneralVocab, PSVAEModel=PSVAEModel)
    return ns


class RefTokenizer:
    """Stand-in for data.mol_bpe.Tokenizer (src/data/mol_bpe.py:156-249) without a vocab
    file: `num_pieces` anonymous pieces followed by '<pad>' and '<s>' (:167-170), and the
    reference's own GeneralVocab with the atom '<pad>' special (:172-173)."""

    def __init__(self, general_vocab_cls, num_pieces=300):
        self.idx2piece = ['P%d' % i for i in range(num_pieces)] + ['<pad>', '<s>']
        self.piece2idx = {p: i for i, p in enumerate(self.idx2piece)}
        self.pad, self.end = '<pad>', '<s>'
        self.atom_pad = '<pad>'
        self.chem_vocab = general_vocab_cls(atom_special=[self.atom_pad])

    def idx_to_piece(self, idx):
        return self.idx2piece[idx]

    def piece_to_idx(self, piece):
        return self.piece2idx[piece]

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
