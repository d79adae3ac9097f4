"""Full-size checks (the benchmark workload: 4096 synthetic ZINC-shaped molecules per batch),
through properties that do not need a CPU oracle run of that size:

* collation: structural invariants of the emitted tensors (CSR order and symmetry, the two
  layouts describing the same graph, the in-piece subset, pair labels, and the exact number of
  kept non-bonded pairs, which the uniform stream fixes);
* model: additivity over molecules.  Every loss is a mean over a count (pairs, piece tokens,
  molecules), so count-weighted losses and their gradients of the whole batch equal the sums
  over its two halves when the halves see the same uniforms and the same eps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 4096
KEEP = 0.022


def dev():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a GPU (run with -m "not gpu" on CPU-only machines)')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def workload():
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    packed = PackedMolecules.from_step1(synthetic_step1(B, seed=2021, shape='zinc'))
    dpacked = packed.to_device(dev())
    uniforms = torch.rand(int(packed.n_rng.sum()) + 8, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    return packed, dpacked, uniforms.to(dev())


def test_collation_invariants_at_batch_4096(workload, tokenizer):
    from ps_vae_amd.data.collate import collate_packed
    packed, dpacked, uniforms = workload
    b = collate_packed(dpacked, np.arange(B), uniforms=uniforms, atom_pad=tokenizer.atom_pad_idx(),
                       piece_pad=tokenizer.pad_idx())
    c = b.compact
    V, N = c.num_nodes, b['x'].shape[1]
    node_off = c.node_off.long()
    assert V == int(node_off[-1]) == int(packed.n_atoms.sum())
    graph_of = torch.repeat_interleave(torch.arange(B, device=dev()), node_off[1:] - node_off[:-1])

    # ---- CSR: monotone row pointers, ascending columns inside a row, neighbours in the same molecule
    row_ptr, col, cls = c.csr.row_ptr.long(), c.csr.col.long(), c.csr.cls.long()
    E = col.numel()
    assert row_ptr[0] == 0 and row_ptr[-1] == E and bool((row_ptr[1:] >= row_ptr[:-1]).all())
    row = torch.repeat_interleave(torch.arange(V, device=dev()), row_ptr[1:] - row_ptr[:-1])
    same_row = row[1:] == row[:-1]
    assert bool((col[1:][same_row] > col[:-1][same_row]).all())
    assert bool((graph_of[row] == graph_of[col]).all()) and bool((row != col).all())
    assert bool(((cls >= 1) & (cls <= 3)).all())
    # symmetric with equal bond classes: {(i, j, c)} == {(j, i, c)}
    fwd_key = torch.sort((row * V + col) * 4 + cls).values
    rev_key = torch.sort((col * V + row) * 4 + cls).values
    assert torch.equal(fwd_key, rev_key)

    # ---- the reference layout describes the same graph
    def to_compact(flat):      # b * N + m  ->  node_off[b] + m
        return node_off[flat // N] + flat % N
    ei = b['edge_index']
    assert torch.equal(to_compact(ei[0]), row) and torch.equal(to_compact(ei[1]), col)
    assert torch.equal(b['edge_attr'], torch.nn.functional.one_hot(cls, 4))
    real = ~b['atom_mask']
    assert int(real.sum()) == V
    for ref_key, cmp_t in (('x', c.cx), ('x_pieces', c.cxp), ('x_pos', c.cxpos)):
        assert torch.equal(b[ref_key][real], cmp_t.long())
    assert bool((b['x'][~real] == tokenizer.atom_pad_idx()).all())
    assert bool((b['x_pieces'][~real] == tokenizer.pad_idx()).all())

    # ---- in-piece bonds = the bonds whose two atoms sit in the same piece (same position id)
    pos = c.cxpos.long()
    inside = pos[row] == pos[col]
    assert torch.equal(b['in_piece_edge_idx'], inside.nonzero().flatten())
    in_ptr, in_col, in_cls = c.csr_in.row_ptr.long(), c.csr_in.col.long(), c.csr_in.cls.long()
    in_row = torch.repeat_interleave(torch.arange(V, device=dev()), in_ptr[1:] - in_ptr[:-1])
    assert torch.equal(in_row, row[inside]) and torch.equal(in_col, col[inside]) and torch.equal(in_cls, cls[inside])

    # ---- candidate pairs: cross-piece, same molecule, row-major order, labelled with the bond class
    pairs, golden = c.pairs, c.golden.long()
    src, dst = pairs.src.long(), pairs.dst.long()
    P = src.numel()
    assert int(b['edge_select'].sum()) == P and torch.equal(b['golden_edge'], golden)
    assert torch.equal(graph_of[src], pairs.graph.long()) and torch.equal(graph_of[dst], pairs.graph.long())
    assert bool((pos[src] != pos[dst]).all())
    pair_key = src * V + dst
    assert bool((pair_key[1:] > pair_key[:-1]).all())                     # row-major, no duplicates
    sel = b['edge_select'].nonzero()                                       # (b, m, n) in row-major order
    assert torch.equal(node_off[sel[:, 0]] + sel[:, 1], src) and torch.equal(node_off[sel[:, 0]] + sel[:, 2], dst)
    edge_key = row * V + col                                               # ascending (CSR order)
    at = torch.searchsorted(edge_key, pair_key).clamp(max=E - 1)
    bonded = edge_key[at] == pair_key
    assert torch.equal(golden, torch.where(bonded, cls[at], torch.zeros_like(golden)))
    assert int(bonded.sum()) == int((~inside).sum())                      # every cross-piece bond is a pair
    # non-bonded candidates draw one uniform each, in order: the kept count is fixed by the stream
    R = b.num_uniforms
    assert R == int(packed.n_rng.sum())
    assert int((~bonded).sum()) == int((uniforms[:R] < KEEP).sum())


def _weighted_sums(model, batch, eps, pad):
    """Count-weighted losses of one batch: each term is a sum over molecules."""
    pl, (rec, piece, edge, kl) = model(batch, eps=eps)
    n_mol = batch.batch_size
    n_tok = (batch.pieces[:, 1:] != pad).sum()
    n_pair = len(batch.compact.pairs)
    n_prop = n_mol * len(model.config['selected_properties'])
    return torch.stack([edge * n_pair, piece * n_tok, kl * n_mol, pl * n_prop])


def test_losses_and_gradients_add_over_half_batches(workload, tokenizer):
    from ps_vae_amd.data.collate import collate_packed, plan
    from ps_vae_amd.pl_models import PSVAEModel
    from ps_vae_amd.utils.nn_utils import build_config, default_args
    packed, dpacked, uniforms = workload
    torch.manual_seed(11)
    model = PSVAEModel(build_config(default_args(), tokenizer), tokenizer).to(dev())
    eps = torch.randn(B, model.decoder.latent_dim, generator=torch.Generator().manual_seed(5)).to(dev())
    half = B // 2
    used_first = plan(packed, np.arange(half))['R']
    kw = dict(atom_pad=tokenizer.atom_pad_idx(), piece_pad=tokenizer.pad_idx(), reference_layout=False)
    parts = [(np.arange(B), uniforms, eps),
             (np.arange(half), uniforms, eps[:half]),
             (np.arange(half, B), uniforms[used_first:], eps[half:])]
    sums, grads, pairs = [], [], []
    for ids, u, e in parts:
        batch = collate_packed(dpacked, ids, uniforms=u, **kw)
        model.zero_grad(set_to_none=True)
        s = _weighted_sums(model, batch, e, tokenizer.pad_idx())
        s.sum().backward()
        sums.append(s.detach().double().cpu())
        pairs.append(len(batch.compact.pairs))
        grads.append({k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None})
    assert pairs[0] == pairs[1] + pairs[2]
    np.testing.assert_allclose((sums[1] + sums[2]).numpy(), sums[0].numpy(), rtol=2e-6)
    assert set(grads[0]) == set(grads[1]) == set(grads[2])
    worst = (0.0, '')
    for k, g in grads[0].items():
        scale = g.abs().max().item()
        err = (grads[1][k] + grads[2][k] - g).abs().max().item()
        worst = max(worst, (err / max(scale, 1e-30), k))
        # identical terms summed in a different order (the K chunks of a weight-gradient product are
        # cut from the row count, so the full batch and its halves chain their fp32 partial sums
        # differently): 1e-5 .. 2e-5 of the largest entry on the sums over all 95 k atoms, whatever the
        # chunking; the run-to-run identity of each sum is test_train_step_repeats_bit_for_bit's subject
        assert err <= 2e-5 * scale + 1e-30, f'{k}: {err / max(scale, 1e-30):.2e} of the largest entry'
    print(f'largest additivity defect: {worst[0]:.2e} of the largest entry ({worst[1]})')


def test_train_step_repeats_bit_for_bit(workload, tokenizer):
    """The same 4096-molecule batch, uniforms and eps through forward + backward three times, two
    streams on: every loss and every gradient repeats bit for bit.  No kernel on the path uses
    atomics or a timing-dependent order of summation, so any difference is a race (a tile read
    before its load had landed showed up exactly like this, as a 1e-5 wobble of one gradient)."""
    from ps_vae_amd.data.collate import collate_packed
    from ps_vae_amd.pl_models import PSVAEModel
    from ps_vae_amd.utils.nn_utils import build_config, default_args
    packed, dpacked, uniforms = workload
    torch.manual_seed(13)
    model = PSVAEModel(build_config(default_args(), tokenizer), tokenizer).to(dev())
    assert model.two_streams
    eps = torch.randn(B, model.decoder.latent_dim, generator=torch.Generator().manual_seed(6)).to(dev())
    runs = []
    for _ in range(3):
        batch = collate_packed(dpacked, np.arange(B), uniforms=uniforms, atom_pad=tokenizer.atom_pad_idx(),
                               piece_pad=tokenizer.pad_idx(), reference_layout=False)
        model.zero_grad(set_to_none=True)
        pl, (rec, piece, edge, kl) = model(batch, eps=eps)
        model.weighted_loss(pl, rec, kl).backward()
        torch.cuda.synchronize()
        runs.append((torch.stack([pl, rec, piece, edge, kl]).detach().clone(),
                     {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    for losses, grads in runs[1:]:
        assert torch.equal(losses, runs[0][0]), (losses, runs[0][0])
        for k, g in grads.items():
            assert torch.equal(g, runs[0][1][k]), f'{k} differs between two runs of the same step'
