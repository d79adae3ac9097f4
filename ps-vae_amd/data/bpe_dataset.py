"""BPEMolDataset / get_dataloader with the reference's names
(src/data/bpe_dataset.py:19-187); collation runs on the GPU.

The dataset holds the molecules in the packed layout (data/packed.py) on the device.  It is built
from a list of process_step1 dicts, a packed .npz file, the reference's processed_data.pkl, or -
like the reference, :157-171 - from a SMILES file through process_step1 (data.mol_bpe: the same
tokenisation on plain graphs instead of RDKit molecules; the three RDKit property scores are
not computed and read as zeros unless supplied)."""
import os
import random

import numpy as np
import torch

from .collate import collate_packed
from .packed import PackedMolecules, load_step1_pickle


class BPEMolDataset(torch.utils.data.Dataset):
    def __init__(self, fname, tokenizer, device='cuda', data=None):
        """fname: a packed .npz (PackedMolecules.save), or — as in the reference, :18-26 — the
        path of a SMILES file whose directory holds the dataset cache 'processed_data.pkl' (a
        torch-saved list of process_step1 dicts, read with weights_only=True).  The packed
        form of that cache is kept beside it as 'processed_data.npz' and preferred when present.
        `data` passes a list of process_step1 dicts (or a PackedMolecules) directly.  With neither
        cache present the SMILES file itself is processed (process, :157-171) and the packed cache
        written beside it."""
        super(BPEMolDataset, self).__init__()
        self.tokenizer = tokenizer
        self.device = torch.device(device)
        if data is None:
            if fname.endswith('.npz'):
                self.packed = PackedMolecules.load(fname)
            else:
                root, _ = os.path.split(fname)
                cache, pkl = os.path.join(root, 'processed_data.npz'), os.path.join(root, 'processed_data.pkl')
                if os.path.exists(cache):
                    self.packed = PackedMolecules.load(cache)
                elif os.path.exists(pkl):
                    self.packed = PackedMolecules.from_step1(load_step1_pickle(pkl))
                    try:
                        self.packed.save(cache)
                    except OSError:
                        pass
                else:
                    self.packed = PackedMolecules.from_step1(self.process(fname))
                    try:
                        self.packed.save(cache)
                    except OSError:
                        pass
        if data is not None:
            self.packed = data if isinstance(data, PackedMolecules) else PackedMolecules.from_step1(data)
        self._check_ids()
        self.dpacked = self.packed.to_device(self.device)

    def _check_ids(self):
        """Every atom type, piece id and bond class must index the tokenizer's tables: the
        reference's nn.Embedding raises on such an id, the kernels here clamp it, so it is refused
        once, on the host, when the dataset is built."""
        p, tok = self.packed, self.tokenizer
        checks = (('atom type', p.atom_type, tok.num_atom_type()), ('piece id', p.piece_id, tok.num_piece_type()),
                  ('bond class', p.bond_cls, tok.chem_vocab.num_bond_type()))
        for what, arr, n in checks:
            if arr.size and (int(arr.min()) < 0 or int(arr.max()) >= n):
                raise ValueError(f'{what} outside [0, {n}): {int(arr.min())} .. {int(arr.max())}')

    @staticmethod
    def process_step1(mol, tokenizer, props=None):
        """Reference :30-54: a molecule (SMILES or utils.molgraph.Mol) to the raw dict
        {x, edge_index, edge_attr, props, pieces, groups}."""
        from . import mol_bpe
        return mol_bpe.process_step1(mol, tokenizer, props)

    def process(self, fname):
        """Reference :157-171: one process_step1 dict per readable SMILES line of `fname`."""
        from ..utils import molgraph
        data_list = []
        with open(fname, 'r') as fin:
            for line in fin:
                mol = molgraph.smiles2molecule(line.strip())
                if mol is not None and len(mol):
                    data_list.append(self.process_step1(mol, self.tokenizer))
        return data_list

    @staticmethod
    def process_step2(data, tokenizer):
        """Reference :56-84 expands a molecule to dense n x n Python lists.  Here that work is
        deferred to the collation kernel: this returns the O(n) per-atom fields and keeps the
        raw dict for process_step3."""
        n = len(data['x'])
        x_pieces, x_pos = [0] * n, [0] * n
        for pos, group in enumerate(data['groups']):
            for aid in group:
                x_pieces[aid] = data['pieces'][pos]
                x_pos[aid] = pos
        return {'x': data['x'], 'props': data['props'], 'pieces': data['pieces'], 'x_pieces': x_pieces,
                'x_pos': x_pos, 'step1': data}

    @staticmethod
    def process_step3(data_list, tokenizer, device='cuda', uniforms=None):
        """Collate process_step2 results (reference :86-155).  The acceptance draws come from
        Python's `random.random()` in the reference's order, so under the same random.seed
        this returns the reference's batch bit for bit; pass `uniforms` (float64 device
        tensor) to supply them instead."""
        packed = PackedMolecules.from_step1([d['step1'] for d in data_list])
        dpacked = packed.to_device(device)
        if uniforms is None:
            need = int(packed.n_rng.sum())
            uniforms = torch.tensor([random.random() for _ in range(need)] or [0.0], dtype=torch.float64).to(device)
        return collate_packed(dpacked, np.arange(len(packed)), uniforms=uniforms,
                              atom_pad=tokenizer.atom_pad_idx(), piece_pad=tokenizer.pad_idx(),
                              num_bond_type=tokenizer.chem_vocab.num_bond_type())

    def collate_ids(self, ids, uniforms=None, reference_layout=False, generator=None):
        """Collate molecules `ids` of this dataset straight from the packed device arrays."""
        return collate_packed(self.dpacked, ids, uniforms=uniforms, atom_pad=self.tokenizer.atom_pad_idx(),
                              piece_pad=self.tokenizer.pad_idx(),
                              num_bond_type=self.tokenizer.chem_vocab.num_bond_type(),
                              reference_layout=reference_layout, generator=generator)

    def __getitem__(self, idx):
        return int(idx)

    def __len__(self):
        return len(self.packed)

    def collate_fn(self, data_list):
        return self.collate_ids(np.asarray(data_list, dtype=np.int64), reference_layout=True)


class _Loader:
    """Batches of molecule ids collated on the GPU (the DataLoader + worker processes of
    the reference, :184-187, are not needed: there is no per-sample Python work left).

    world > 1: this rank's share of every epoch, as torch's DistributedSampler deals it
    (src/guacamol_exps/goal_directed/train.py:271): all ranks build the same order (the same
    seed, advanced by the epoch count), `pad` repeats molecules from its start until it divides
    by the world size, and rank r takes order[r::world] - disjoint shares and the same number of
    steps on every rank, which the per-step gradient all-reduce needs.  pad=False (evaluation:
    no collective inside a step) leaves the order as it is, so every molecule is seen once."""

    def __init__(self, dataset, batch_size, shuffle, reference_layout=True, seed=None, rank=0, world=1, pad=True):
        if not 0 <= rank < world:
            raise ValueError(f'rank {rank} outside a world of {world}')
        if world > 1 and shuffle and seed is None:
            raise ValueError('a sharded, shuffled loader needs a seed that every rank shares')
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.reference_layout = reference_layout
        self.seed, self.rank, self.world, self.pad = seed, rank, world, pad
        self.epoch = 0
        self.rng = np.random.default_rng(seed)

    def _share(self, n):
        if self.world == 1:
            return n
        return (n + self.world - 1) // self.world if self.pad else len(range(self.rank, n, self.world))

    def __len__(self):
        return (self._share(len(self.dataset)) + self.batch_size - 1) // self.batch_size

    def order(self):
        """Molecule ids of the next epoch, in visiting order (this rank's share)."""
        n = len(self.dataset)
        if self.world == 1:
            return self.rng.permutation(n) if self.shuffle else np.arange(n)
        order = np.random.default_rng(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        if self.pad and n % self.world and n:
            order = np.concatenate([order, np.resize(order, self.world - n % self.world)])
        return order[self.rank::self.world]

    def __iter__(self):
        order = self.order()
        self.epoch += 1
        for i in range(0, len(order), self.batch_size):
            yield self.dataset.collate_ids(order[i:i + self.batch_size], reference_layout=self.reference_layout)


def get_dataloader(fname, tokenizer, batch_size, shuffle=False, num_workers=4, device='cuda', data=None,
                   reference_layout=True, seed=None, rank=0, world=1, pad=True):
    """reference_layout=False yields batches that carry only the compact layout the kernels
    read (no padded [B,N] tensors, no host stop for the pair count): what training uses.
    rank / world / pad: see _Loader."""
    dataset = BPEMolDataset(fname, tokenizer, device=device, data=data)
    return _Loader(dataset, batch_size, shuffle, reference_layout=reference_layout, seed=seed, rank=rank, world=world,
                   pad=pad)
