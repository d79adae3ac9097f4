"""The multi-process gradient exchange (ps_vae_amd.dist.FlatGradSync) on CPU: world_size 2,
gloo, driving the CPU oracle model.  The averaged gradient must equal the mean of the two
single-rank gradients, parameters without a gradient must not stall the exchange, and both
ranks must hold identical parameters after the optimiser step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SMALL = dict(latent_dim=8, predictor_hidden_dim=16, node_hidden_dim=36, graph_embedding_dim=40,
             atom_embedding_dim=6, piece_embedding_dim=12, pos_embedding_dim=6, piece_hidden_dim=20)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank_batch(rank, tokenizer):
    import ps_vae_amd  # noqa: F401
    from oracle import collate_ref
    from ps_vae_amd.data.synthetic import synthetic_step1
    step1 = synthetic_step1(6, seed=2021 + rank, shape='zinc')
    uniforms = np.random.default_rng(rank).random(6 * 700)
    batch, _ = collate_ref.collate(step1, uniforms, piece_pad=tokenizer.pad_idx())
    eps = torch.randn(6, SMALL['latent_dim'], generator=torch.Generator().manual_seed(10 + rank))
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}, eps


def _make_model(tokenizer):
    from oracle import model_ref
    torch.manual_seed(0)
    return model_ref.PSVAEModel(model_ref.default_config(tokenizer, **SMALL), tokenizer)


def _local_grads(model, batch, eps):
    model.zero_grad(set_to_none=True)
    pl, (rec, _, _, kl) = model(batch, eps)
    model.weighted_loss(pl, rec, kl).backward()
    return {k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import ps_vae_amd  # noqa: F401
    from conftest import Tokenizer
    from ps_vae_amd.dist import FlatGradSync
    tok = Tokenizer()
    model = _make_model(tok)
    sync = FlatGradSync(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    batch, eps = _rank_batch(rank, tok)
    sync.zero_grad()
    pl, (rec, _, _, kl) = model(batch, eps)
    model.weighted_loss(pl, rec, kl).backward()
    sync.sync()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    norm = sync.clip_grad_norm_(10.0)
    opt.step()
    torch.save({'grads': grads, 'norm': norm, 'state': model.state_dict()}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_sync_world_size_2(tmp_path, tokenizer):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f'rank{r}.pt'), weights_only=True) for r in range(world)]
    # expected: mean of the single-rank gradients
    model = _make_model(tokenizer)
    locals_ = [_local_grads(model, *_rank_batch(r, tokenizer)) for r in range(world)]
    for k in locals_[0]:
        if locals_[0][k] is None:
            assert torch.count_nonzero(res[0]['grads'][k]) == 0      # never-used parameters stay zero
            continue
        want = (locals_[0][k] + locals_[1][k]) / world
        for r in range(world):
            torch.testing.assert_close(res[r]['grads'][k], want, rtol=1e-6, atol=1e-9)
    assert torch.equal(res[0]['norm'], res[1]['norm'])
    for k in res[0]['state']:
        assert torch.equal(res[0]['state'][k], res[1]['state'][k]), k


def test_flat_grad_sync_single_process_matches_plain_training(tokenizer):
    """Without a process group the helper must reproduce clip_grad_norm_ + Adam exactly."""
    import ps_vae_amd  # noqa: F401
    from ps_vae_amd.dist import FlatGradSync
    batch, eps = _rank_batch(0, tokenizer)
    a, b = _make_model(tokenizer), _make_model(tokenizer)
    opt_a = torch.optim.Adam(a.parameters(), lr=1e-3)
    pl, (rec, _, _, kl) = a(batch, eps)
    a.weighted_loss(pl, rec, kl).backward()
    na = torch.nn.utils.clip_grad_norm_(a.parameters(), 0.05)
    opt_a.step()
    sync = FlatGradSync(b.parameters())
    opt_b = torch.optim.Adam(b.parameters(), lr=1e-3)
    sync.zero_grad()
    pl, (rec, _, _, kl) = b(batch, eps)
    b.weighted_loss(pl, rec, kl).backward()
    sync.sync()
    nb = sync.clip_grad_norm_(0.05)
    opt_b.step()
    torch.testing.assert_close(na, nb, rtol=1e-5, atol=0)   # one flat norm vs a norm of per-tensor norms
    skip = {'decoder.graph_embedding.linear.weight', 'decoder.graph_embedding.linear.bias'}
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if k not in skip:
            torch.testing.assert_close(pa, pb, rtol=1e-5, atol=1e-7, msg=k)
