"""Host-side logic that needs no GPU: the packed molecule format, the batch planner, config
builders and the model's schedule helpers."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz, sub
from oracle import collate_ref, model_ref

import ps_vae_amd  # noqa: F401
from ps_vae_amd.data.collate import plan
from ps_vae_amd.data.packed import PackedMolecules
from ps_vae_amd.data.synthetic import synthetic_step1


def all_cases():
    with open(os.path.join(GOLDEN, 'collate_cases.json')) as f:
        return json.load(f)


def test_packed_round_trip_and_counts(tokenizer):
    """from_step1 -> to_step1 collates to the same batch, and the per-molecule counts the
    kernels' offsets are built from equal what the reference's loops produce."""
    gold_all = load_npz('collate_cases.npz')
    for name, step1 in all_cases().items():
        gold = sub(gold_all, name)
        packed = PackedMolecules.from_step1(step1)
        back = [packed.to_step1(i) for i in range(len(packed))]
        batch, used = collate_ref.collate(back, gold['uniforms'], piece_pad=tokenizer.pad_idx())
        for k in ('x', 'x_pieces', 'x_pos', 'edge_index', 'edge_attr', 'edge_select', 'golden_edge',
                  'in_piece_edge_idx', 'pieces'):
            if gold[k].size == 0:
                assert batch[k].size == 0
                continue
            assert np.array_equal(batch[k], gold[k]), (name, k)
        p = plan(packed, np.arange(len(packed)))
        assert p['E'] == gold['edge_index'].shape[1], name
        assert p['Ein'] == len(gold['in_piece_edge_idx']), name
        assert p['R'] == len(gold['uniforms']) == used, name
        assert p['N'] == gold['x'].shape[1] and p['S'] == gold['pieces'].shape[1], name
        assert p['V'] == int((~gold['atom_mask']).sum()), name


def test_packed_counts_per_molecule_on_synthetic(tokenizer):
    step1 = synthetic_step1(64, seed=5, shape='zinc') + synthetic_step1(32, seed=6, shape='qm9')
    packed = PackedMolecules.from_step1(step1)
    big = np.random.default_rng(0).random(5000)
    for i, d in enumerate(step1):
        batch, used = collate_ref.collate([d], big, piece_pad=tokenizer.pad_idx())
        assert packed.n_edge[i] == batch['edge_index'].shape[1]
        assert packed.n_inpiece[i] == len(batch['in_piece_edge_idx'])
        assert packed.n_rng[i] == used


def test_packed_select_and_save(tmp_path):
    step1 = synthetic_step1(10, seed=8, shape='zinc')
    packed = PackedMolecules.from_step1(step1)
    ids = [7, 2, 2, 9]
    sel = packed.select(ids)
    for j, i in enumerate(ids):
        assert sel.to_step1(j) == packed.to_step1(i)
    path = str(tmp_path / 'p.npz')
    packed.save(path)
    again = PackedMolecules.load(path)
    assert all(again.to_step1(i) == packed.to_step1(i) for i in range(len(packed)))


def test_packed_rejects_bad_input():
    end = 301
    with pytest.raises(ValueError):
        PackedMolecules.from_step1([{'x': [2, 2], 'edge_index': [[0, 5]], 'edge_attr': [1], 'props': [0, 0, 0],
                                     'pieces': [end, 1, end], 'groups': [[], [0, 1], []]}])
    with pytest.raises(ValueError):
        PackedMolecules.from_step1([{'x': [2, 2], 'edge_index': [], 'edge_attr': [], 'props': [0, 0, 0],
                                     'pieces': [end, 1, 2, end], 'groups': [[], [0, 1], [1], []]}])


def test_duplicate_and_none_bonds_resolve_like_the_dense_assignment(tokenizer):
    """ad_mat[a][b] = ad_mat[b][a] = attr in list order (bpe_dataset.py:61-63): the last
    entry for a pair wins, a None (0) entry erases the bond."""
    end = 301
    d = {'x': [2, 2, 7], 'edge_index': [[0, 1], [1, 0], [1, 2], [2, 1]], 'edge_attr': [1, 2, 3, 0],
         'props': [0.1, 0.2, 0.3], 'pieces': [end, 4, 5, end], 'groups': [[], [0], [1, 2], []]}
    want, used = collate_ref.collate([d], np.full(20, 0.5), piece_pad=tokenizer.pad_idx())
    packed = PackedMolecules.from_step1([d])
    got, used2 = collate_ref.collate([packed.to_step1(0)], np.full(20, 0.5), piece_pad=tokenizer.pad_idx())
    assert used == used2 == packed.n_rng[0]
    for k in ('edge_index', 'edge_attr', 'edge_select', 'golden_edge'):
        assert np.array_equal(want[k], got[k])
    assert want['edge_attr'].argmax(1).tolist() == [2, 2]


def test_config_builders_and_beta(tokenizer):
    from ps_vae_amd.pl_models import PSVAEModel
    from ps_vae_amd.utils.nn_utils import build_config, default_args
    cfg = build_config(default_args(), tokenizer)
    want = model_ref.default_config(tokenizer)
    for k in ('lr', 'alpha', 'beta', 'max_beta', 'step_beta', 'kl_warmup', 'kl_anneal_iter',
              'selected_properties', 'predictor', 'encoder'):
        assert cfg[k] == want[k], k
    assert {k: v for k, v in cfg['vae_piece_decoder'].items() if k != 'tokenizer'} == \
           {k: v for k, v in want['vae_piece_decoder'].items() if k != 'tokenizer'}
    model = PSVAEModel(cfg, tokenizer)
    gold = load_npz('beta_table.npz')
    for s in (0, 1, 999, 1000, 2500, 4000, 5999):
        model.global_step = s
        assert model.cal_beta() == gold['train_sh'][s]
    opt, sched = model.configure_optimizers()
    assert isinstance(opt[0], torch.optim.Adam) and opt[0].param_groups[0]['lr'] == 1e-3
    sched[0].step()
    assert opt[0].param_groups[0]['lr'] == pytest.approx(5e-4)     # LambdaLR 1/(epoch+1)


def test_tokenizer_vocab_file(tmp_path):
    from ps_vae_amd.utils.tokenizer import Tokenizer
    path = tmp_path / 'vocab.txt'
    path.write_text('C\t1\t100\nCC\t2\t50\nc1ccccc1\t6\t20\n')
    tok = Tokenizer(str(path))
    assert tok.num_piece_type() == 5 and tok.pad_idx() == 3 and tok.end_idx() == 4
    assert tok.piece_to_idx('CC') == 1 and tok.idx_to_piece(2) == 'c1ccccc1'
    assert tok.num_atom_type() == 11 and tok.atom_pad_idx() == 10
    assert tok.chem_vocab.bond_to_idx(None) == 0 and tok.chem_vocab.num_bond_type() == 4


def test_cnt_atom_and_piece_atom_counts(tmp_path):
    """cnt_atom counts the single characters that are MAX_VALENCE keys
    (src/utils/chem_utils.py:158-163); the tokenizer exposes it per piece id."""
    from ps_vae_amd.utils.chem_vocab import cnt_atom
    from ps_vae_amd.utils.tokenizer import Tokenizer
    assert cnt_atom('C1=CC=CC=C1') == 6
    assert cnt_atom('CCl') == 2 and cnt_atom('BrC') == 2 and cnt_atom('<s>') == 0 and cnt_atom('C[NH3+]') == 2
    vocab = tmp_path / 'vocab.txt'
    vocab.write_text('C\t1\t10\nCC\t2\t5\nC1CC1\t3\t2\n')
    tok = Tokenizer(str(vocab))
    counts = tok.piece_atom_counts()
    assert counts[:3] == [1, 2, 3] and counts[tok.pad_idx()] == 0 and counts[tok.end_idx()] == 0



def test_processed_data_pkl_round_trip(tmp_path, tokenizer):
    """The reference's dataset cache ('processed_data.pkl': torch.save of the process_step1
    dicts, src/data/bpe_dataset.py:23-26, 170) -> BPEMolDataset -> packed arrays, through the
    converter script in both directions."""
    import subprocess
    import sys
    from ps_vae_amd.data.bpe_dataset import BPEMolDataset
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    step1 = synthetic_step1(40, seed=5, shape='zinc') + synthetic_step1(10, seed=6, shape='qm9')
    want = PackedMolecules.from_step1(step1)
    torch.save(step1, tmp_path / 'processed_data.pkl')
    (tmp_path / 'train.txt').write_text('C\n')
    ds = BPEMolDataset(str(tmp_path / 'train.txt'), tokenizer, device='cpu')
    assert len(ds) == 50
    for k in ('atom_off', 'atom_type', 'atom_pos', 'bond_off', 'bond_a', 'bond_b', 'bond_cls', 'piece_off',
              'piece_id', 'n_edge', 'n_inpiece', 'n_rng', 'props'):
        assert np.array_equal(getattr(ds.packed, k), getattr(want, k)), k
    assert (tmp_path / 'processed_data.npz').exists()            # packed cache written beside the pickle
    ds2 = BPEMolDataset(str(tmp_path / 'train.txt'), tokenizer, device='cpu')
    assert np.array_equal(ds2.packed.atom_type, want.atom_type)
    # the converter script, both directions
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'pkl_to_packed.py')
    subprocess.run([sys.executable, tool, str(tmp_path / 'processed_data.pkl'), str(tmp_path / 'a.npz')], check=True)
    subprocess.run([sys.executable, tool, '--to-pkl', str(tmp_path / 'a.npz'), str(tmp_path / 'back.pkl')], check=True)
    back = torch.load(tmp_path / 'back.pkl', weights_only=True)
    again = PackedMolecules.from_step1(back)
    assert np.array_equal(again.atom_pos, want.atom_pos) and np.array_equal(again.bond_cls, want.bond_cls)
    assert np.array_equal(again.piece_id, want.piece_id) and np.allclose(again.props, want.props)
    with pytest.raises(FileNotFoundError):
        BPEMolDataset(str(tmp_path / 'nowhere' / 'train.txt'), tokenizer, device='cpu')


def test_dataset_refuses_ids_outside_the_tokenizer_tables(tokenizer):
    """An atom type or piece id that indexes no embedding row is refused when the dataset is
    built (nn.Embedding in the reference raises on it at the first forward)."""
    from ps_vae_amd.data.bpe_dataset import BPEMolDataset
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    mols = synthetic_step1(3, seed=2, shape='zinc')
    ok = BPEMolDataset.__new__(BPEMolDataset)
    ok.packed, ok.tokenizer = PackedMolecules.from_step1(mols), tokenizer
    ok._check_ids()
    for field, bad in (('x', tokenizer.num_atom_type()), ('pieces', tokenizer.num_piece_type()), ('x', -1)):
        broken = [dict(m) for m in mols]
        broken[1][field] = list(broken[1][field])
        broken[1][field][0] = bad
        ds = BPEMolDataset.__new__(BPEMolDataset)
        ds.packed, ds.tokenizer = PackedMolecules.from_step1(broken), tokenizer
        with pytest.raises(ValueError):
            ds._check_ids()
