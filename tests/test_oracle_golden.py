"""The CPU oracle (oracle/*.py) against the golden vectors produced by the reference's own
code (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz, sub
from oracle import collate_ref, model_ref

INT_KEYS = ['x', 'x_pieces', 'x_pos', 'atom_mask', 'pieces', 'edge_index', 'edge_attr', 'edge_select',
            'golden_edge', 'in_piece_edge_idx']


def collate_cases():
    with open(os.path.join(GOLDEN, 'collate_cases.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('name', ['single_atom', 'one_piece_only', 'no_bond_only', 'mixed_small',
                                  'zinc_ragged', 'qm9_ragged', 'max_size_38'])
def test_collate_bit_exact(name, tokenizer):
    step1 = collate_cases()[name]
    gold = sub(load_npz('collate_cases.npz'), name)
    batch, used = collate_ref.collate(step1, gold['uniforms'], piece_pad=tokenizer.pad_idx())
    assert used == len(gold['uniforms'])
    for k in INT_KEYS:
        want = gold[k]
        got = batch[k]
        if k == 'edge_index' and want.shape[1] == 0:
            assert got.shape == (2, 0)
            continue
        assert got.shape == want.shape, k
        assert got.dtype == want.dtype, k
        assert np.array_equal(got, want), k
    assert np.array_equal(batch['props'], gold['props'].astype(np.float32).reshape(len(step1), -1))


def to_torch_batch(npb):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in npb.items()}


def small_model(tokenizer, gold):
    cfg = model_ref.default_config(tokenizer, latent_dim=8, predictor_hidden_dim=16, node_hidden_dim=36,
                                   graph_embedding_dim=40, atom_embedding_dim=6, piece_embedding_dim=12,
                                   pos_embedding_dim=6, piece_hidden_dim=20)
    model = model_ref.PSVAEModel(cfg, tokenizer)
    state = {k: torch.from_numpy(v) for k, v in sub(gold, 'state').items()}
    missing = model.load_state_dict(state, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model, cfg


def test_model_small_forward_backward_step(tokenizer):
    gold = load_npz('model_small.npz')
    model, cfg = small_model(tokenizer, gold)
    batch = to_torch_batch(sub(gold, 'batch'))
    eps = torch.from_numpy(gold['eps'])

    with torch.no_grad():
        x = model.decoder.embed_atom(batch['x'], batch['x_pieces'], batch['x_pos'])
        assert torch.equal(x, torch.from_numpy(gold['act/embed_atom']))
        B, N, D = x.shape
        last, all_x = model.encoder.embed_node(x.view(-1, D), batch['edge_index'], batch['edge_attr'])
        np.testing.assert_allclose(last.numpy(), gold['act/enc_last'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(all_x.numpy(), gold['act/enc_all_x'], rtol=1e-6, atol=1e-6)
        gids = torch.repeat_interleave(torch.arange(B), N)
        genc = model.encoder.embed_graph(all_x, gids, batch['atom_mask'].flatten())
        np.testing.assert_allclose(genc.numpy(), gold['act/graph_embedding'], rtol=1e-6, atol=1e-6)
        z, kl = model.decoder.rsample(genc, eps)
        np.testing.assert_allclose(z.numpy(), gold['act/z'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(kl.numpy(), gold['act/kl'], rtol=1e-6)

    pred_loss, (losses, accus) = model(batch, eps, return_accu=True)
    rec, piece, edge, kl = losses
    loss = model.weighted_loss(pred_loss, rec, kl)
    for name, val in [('total', loss), ('pred', pred_loss), ('rec', rec), ('piece', piece), ('edge', edge), ('kl', kl)]:
        np.testing.assert_allclose(val.item(), gold[f'loss/{name}'], rtol=1e-6, err_msg=name)
    assert accus[0] == pytest.approx(float(gold['accu/piece']))
    assert accus[1] == pytest.approx(float(gold['accu/edge']))
    loss.backward()
    no_grad = set(gold['no_grad_params'].tolist())
    for k, p in model.named_parameters():
        if k in no_grad:
            assert p.grad is None
            continue
        np.testing.assert_allclose(p.grad.numpy(), gold[f'grad/{k}'], rtol=1e-4, atol=1e-7, err_msg=k)
    opt = torch.optim.Adam(model.parameters(), lr=cfg['lr'])
    total = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    np.testing.assert_allclose(total.item(), gold['grad_total_norm'], rtol=1e-5)
    opt.step()
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), gold[f'after_step/{k}'], rtol=1e-5, atol=1e-6, err_msg=k)


def test_model_default_width_from_seed(tokenizer):
    """Default widths (H=300, G=400, L=56): weights are re-created from the seed in the same
    construction order as the reference, outputs compared with the stored ones."""
    gold = load_npz('model_default.npz')
    cfg = model_ref.default_config(tokenizer)
    torch.manual_seed(int(gold['seed']))
    model = model_ref.PSVAEModel(cfg, tokenizer)
    batch = to_torch_batch(sub(gold, 'batch'))
    eps = torch.from_numpy(gold['eps'])
    pred_loss, (rec, piece, edge, kl) = model(batch, eps)
    loss = model.weighted_loss(pred_loss, rec, kl)
    for name, val in [('total', loss), ('pred', pred_loss), ('rec', rec), ('piece', piece), ('edge', edge), ('kl', kl)]:
        np.testing.assert_allclose(val.item(), gold[f'loss/{name}'], rtol=2e-6, err_msg=name)
    loss.backward()
    for k, p in model.named_parameters():
        if p.grad is None:
            assert k in set(gold['no_grad_params'].tolist())
            continue
        np.testing.assert_allclose(p.grad.norm().item(), gold[f'gradnorm/{k}'], rtol=1e-4, err_msg=k)
        # element-wise: every 97th entry of every gradient, within 1e-5 of the tensor's largest entry
        got = p.grad.reshape(-1)[::97].numpy()
        assert np.abs(got - gold[f'gradsample/{k}']).max() <= 1e-5 * float(gold[f'gradmax/{k}']), k
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params == 1891727      # encoder 722 800 + decoder 1 116 925 + predictor 52 002


def test_cal_beta_table(tokenizer):
    gold = load_npz('beta_table.npz')
    for name in ('train_sh', 'warmup'):
        warm, anneal, max_beta, step_beta, beta0 = gold[name + '/config']
        cfg = {'kl_warmup': int(warm), 'kl_anneal_iter': int(anneal), 'max_beta': float(max_beta),
               'step_beta': float(step_beta), 'beta': float(beta0)}
        got = np.asarray([model_ref.cal_beta(int(s), cfg) for s in gold['steps']], dtype=np.float64)
        assert np.array_equal(got, gold[name])


def test_infer_links(tokenizer):
    gold = load_npz('infer_links.npz')
    model, _ = small_model(tokenizer, gold)
    th = float(gold['add_edge_th'])
    for c in range(int(gold['num_cases'])):
        g = sub(gold, f'case{c}')
        groups, p = [], 0
        for ln in g['groups_len']:
            groups.append(g['groups_flat'][p:p + ln].tolist())
            p += ln
        with torch.no_grad():
            pairs, conf, etype, order = model.decoder.infer_links(
                torch.from_numpy(g['z']), torch.from_numpy(g['x']), torch.from_numpy(g['x_pieces']),
                torch.from_numpy(g['x_pos']), torch.from_numpy(g['edge_index']),
                torch.from_numpy(g['edge_attr']), groups, th)
        assert np.array_equal(pairs.numpy(), g['pairs'])
        np.testing.assert_allclose(conf.numpy(), g['conf'], rtol=1e-5)
        assert np.array_equal(etype.numpy(), g['etype'])
        assert order == g['order'].tolist()


def test_compact_layout_consistency(tokenizer):
    """compact_from_reference is index bookkeeping: expanding it back must give the
    reference batch."""
    step1 = collate_cases()['zinc_ragged']
    gold = sub(load_npz('collate_cases.npz'), 'zinc_ragged')
    batch, _ = collate_ref.collate(step1, gold['uniforms'], piece_pad=tokenizer.pad_idx())
    c = collate_ref.compact_from_reference(batch)
    B, N = batch['x'].shape
    V = c['node_off'][-1]
    graph_of = np.repeat(np.arange(B), np.diff(c['node_off']))
    local = np.arange(V) - c['node_off'][graph_of]
    padded = graph_of * N + local
    src = np.repeat(np.arange(V), np.diff(c['row_ptr']))
    assert np.array_equal(padded[src], batch['edge_index'][0])
    assert np.array_equal(padded[c['col']], batch['edge_index'][1])
    assert np.array_equal(c['cls'], batch['edge_attr'].argmax(1))
    sel = np.zeros_like(batch['edge_select'])
    sel[c['pair_graph'], local[c['pair_src']], local[c['pair_dst']]] = True
    assert np.array_equal(sel, batch['edge_select'])
    assert len(c['in_col']) == len(batch['in_piece_edge_idx'])


def test_assemble_checks_against_reference_answers(tokenizer):
    """valence_check / cycle_check / shortest_path_len of the restatement give the answers the
    reference's own functions gave on the same inputs (tests/golden/assemble_checks.npz)."""
    from oracle import assemble_ref
    g = load_npz('assemble_checks.npz')
    vocab = tokenizer.chem_vocab
    for k in range(len(g['v_ok'])):
        e1 = g['v_edges'][k, 0, :g['v_len'][k, 0]].tolist()
        e2 = g['v_edges'][k, 1, :g['v_len'][k, 1]].tolist()
        got = assemble_ref.valence_check(int(g['v_atoms'][k, 0]), int(g['v_atoms'][k, 1]), e1, e2,
                                         int(g['v_new'][k]), vocab)
        assert bool(got) == bool(g['v_ok'][k]), k
    for k in range(len(g['c_ok'])):
        n = int(g['c_n'][k])
        adj = [[] for _ in range(n)]
        for a, b in g['c_edges'][k]:
            if a >= 0:
                adj[a].append(b), adj[b].append(a)
        i, j = (int(v) for v in g['c_pair'][k])
        assert (assemble_ref.shortest_path_len(i, j, adj) or 0) == int(g['c_len'][k]), k
        assert bool(assemble_ref.cycle_check(i, j, adj)) == bool(g['c_ok'][k]), k


def test_assemble_loop_small_cases(tokenizer):
    """The greedy loop on hand-checked cases: a 6-ring closes, a 4-ring does not, a bond that
    would exceed a valence is refused, blocks merge and the largest survives."""
    from oracle import assemble_ref
    vocab = tokenizer.chem_vocab
    C, O = vocab.atom_to_idx('C'), vocab.atom_to_idx('O')
    # two 3-atom chains C-C-C: joining 2-3 makes a 6-chain, then 0-5 closes a 6-ring, 1-4 would close a 4-ring
    atoms = [C] * 6 + [O]
    bonds = [(0, 1, 1), (1, 2, 1), (3, 4, 1), (4, 5, 1)]
    groups = [[0, 1, 2], [3, 4, 5], [6]]
    cand = [(2, 3), (0, 5), (1, 4), (0, 6)]
    acc, block, keep = assemble_ref.assemble(atoms, bonds, groups, cand, [1, 1, 1, 3], [0, 1, 2, 3], vocab)
    assert acc == [True, True, False, False]        # 0-6 triple: O allows 2
    assert block[:6] == [1] * 6 and block[6] == 2
    assert keep == [True] * 6 + [False]


class _PieceTokenizer:
    """The vocabulary stored with tests/golden/inference_cases.npz."""

    def __init__(self, pieces):
        from ps_vae_amd.utils.chem_vocab import GeneralVocab
        self.idx2piece = list(pieces)
        self.chem_vocab = GeneralVocab(atom_special=['<pad>'])

    def idx_to_piece(self, i):
        return self.idx2piece[i]

    def pad_idx(self):
        return len(self.idx2piece) - 2

    def end_idx(self):
        return len(self.idx2piece) - 1

    def atom_pad_idx(self):
        return self.chem_vocab.atom_to_idx('<pad>')

    def num_piece_type(self):
        return len(self.idx2piece)

    def num_atom_type(self):
        return self.chem_vocab.num_atom_type()


def test_inference_end_to_end_against_reference_cases():
    """The oracle's inference (sampling loop, fragment expansion, link slice, greedy assembly,
    largest block) against what the reference's own VAEPieceDecoder.inference produced on graph
    stand-ins for the RDKit objects (tests/golden/make_golden.py: make_inference)."""
    from oracle import assemble_ref
    from ps_vae_amd.utils.fragments import parse_piece_smiles
    gold = load_npz('inference_cases.npz')
    tok = _PieceTokenizer(gold['pieces'].tolist())
    model, _ = small_model(tok, gold)
    fragments = [([], []) if p.startswith('<') else parse_piece_smiles(p) for p in tok.idx2piece]
    th, max_atoms, temp = float(gold['add_edge_th']), int(gold['max_atom_num']), float(gold['temperature'])
    assert int(gold['num_cases']) >= 40
    for c in range(int(gold['num_cases'])):
        g = sub(gold, f'case{c}')
        with torch.no_grad():
            atoms, bonds = model.decoder.inference(torch.from_numpy(g['z']), max_atoms, th, temp,
                                                   torch.from_numpy(g['uniforms']), fragments, assemble_ref.assemble)
        assert atoms == g['atoms'].tolist(), c
        assert bonds == [tuple(r) for r in g['bonds'].tolist()], c
