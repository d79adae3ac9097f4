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


# ---------------------------------------------------------------- the sharded fit loop, world 2
class _HostDataset:
    """Stands in for BPEMolDataset under _Loader: molecules collated on the host by the oracle;
    remembers which ids every batch held."""

    def __init__(self, step1, tokenizer, seed):
        self.step1, self.tok, self.seen = step1, tokenizer, []
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return len(self.step1)

    def collate_ids(self, ids, reference_layout=True):
        from oracle import collate_ref
        ids = [int(i) for i in ids]
        self.seen.append(ids)
        batch, _ = collate_ref.collate([self.step1[i] for i in ids], self.rng.random(700 * len(ids)),
                                       piece_pad=self.tok.pad_idx())
        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}


def _hooked_model(tokenizer, rank):
    """The oracle model with the hook methods ps_vae_amd.train.fit drives."""
    from oracle import model_ref

    class Hooked(model_ref.PSVAEModel):
        def log(self, name, value):
            self.logged[name] = value.detach() if torch.is_tensor(value) else value

        def _eps(self, batch):
            return torch.randn(batch['x'].shape[0], SMALL['latent_dim'], generator=self.gen)

        def training_step(self, batch, idx):
            pl, (rec, _, _, kl) = self(batch, self._eps(batch))
            return self.weighted_loss(pl, rec, kl)

        def validation_step(self, batch, idx):
            pl, (rec, _, _, kl) = self(batch, torch.zeros(batch['x'].shape[0], SMALL['latent_dim']))
            self.log('val_loss', self.weighted_loss(pl, rec, kl))

        def configure_optimizers(self):
            opt = torch.optim.Adam(self.parameters(), lr=1e-3)
            return [opt], [torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda epoch: 1 / (epoch + 1))]

    torch.manual_seed(0)
    model = Hooked(model_ref.default_config(tokenizer, **SMALL), tokenizer)
    model.logged, model.gen = {}, torch.Generator().manual_seed(100 + rank)
    return model


def _fit_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import ps_vae_amd  # noqa: F401
    from conftest import Tokenizer
    from ps_vae_amd import train
    from ps_vae_amd.data.bpe_dataset import _Loader
    from ps_vae_amd.data.synthetic import synthetic_step1
    got = train.init_distributed(backend='gloo')
    assert got[:2] == (rank, world)
    tok = Tokenizer()
    train_ds = _HostDataset(synthetic_step1(13, seed=5, shape='zinc'), tok, seed=rank)
    valid_ds = _HostDataset(synthetic_step1(5, seed=6, shape='zinc'), tok, seed=50 + rank)
    loaders = (_Loader(train_ds, 3, True, seed=7, rank=rank, world=world),
               _Loader(valid_ds, 2, False, rank=rank, world=world, pad=False))
    assert len(loaders[0]) == 3 and len(loaders[1]) == (2 if rank == 0 else 1)
    model = _hooked_model(tok, rank)
    hist = train.fit(model, *loaders, max_epochs=2, grad_clip=10.0, save_dir=os.path.join(out_dir, 'ckpt'))
    torch.save({'state': model.state_dict(), 'train_seen': train_ds.seen, 'valid_seen': valid_ds.seen,
                'monitor': hist['monitor'], 'steps': hist['global_step'], 'best_path': hist['best_path']},
               os.path.join(out_dir, f'fit{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_fit_world_size_2_shards_the_data_and_keeps_the_ranks_identical(tmp_path):
    """ps_vae_amd.train.fit under a world of 2 (gloo, the CPU oracle model): every epoch deals the
    13 training molecules to the two ranks in disjoint shares of equal length (one molecule repeated
    to make 14), the validation set is split without repeats and its loss reduced over the ranks,
    the parameters stay identical, and only rank 0 writes the checkpoint."""
    world = 2
    mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f'fit{r}.pt'), weights_only=True) for r in range(world)]
    for k in res[0]['state']:
        assert torch.equal(res[0]['state'][k], res[1]['state'][k]), k
    assert res[0]['steps'] == res[1]['steps'] == 6                       # 2 epochs x ceil(7 / 3) steps
    assert res[0]['monitor'] == res[1]['monitor'] and len(res[0]['monitor']) == 2
    per_epoch = [[sum(r['train_seen'][e * 3:(e + 1) * 3], []) for r in res] for e in range(2)]
    for shares in per_epoch:
        assert len(shares[0]) == len(shares[1]) == 7
        both = shares[0] + shares[1]
        assert sorted(set(both)) == list(range(13))                     # every molecule is seen
        assert len(both) - len(set(both)) == 1                          # one repeat pads 13 to 14
    assert per_epoch[0] != per_epoch[1]                                  # reshuffled between epochs
    valid = [sum(r['valid_seen'][:len(r['valid_seen']) // 2], []) for r in res]
    assert sorted(valid[0] + valid[1]) == list(range(5)) and valid[0] == [0, 2, 4]
    files = os.listdir(os.path.join(tmp_path, 'ckpt'))
    assert len(files) == 1 and res[0]['best_path'].endswith(files[0]) and res[1]['best_path'] == res[0]['best_path']


def test_checkpoint_rebuilds_the_model_by_itself(tmp_path):
    """A ModelCheckpoint file holds config and vocabulary: PSVAEModel.load_from_checkpoint rebuilds
    tokenizer and model from the file alone (weights_only load) with equal weights."""
    import ps_vae_amd  # noqa: F401
    from ps_vae_amd import train
    from ps_vae_amd.pl_models import PSVAEModel
    from ps_vae_amd.utils.nn_utils import build_config, default_args
    from ps_vae_amd.utils.tokenizer import Tokenizer
    tokenizer = Tokenizer(num_pieces=40)
    args = default_args(latent_dim=8, node_hidden_dim=36, graph_embedding_dim=40, props=['qed'])
    torch.manual_seed(3)
    model = PSVAEModel(build_config(args, tokenizer), tokenizer)
    model.global_step = 17
    path = train.ModelCheckpoint(str(tmp_path)).on_validation_end(model, 2, 17, 1.25)
    twin = PSVAEModel.load_from_checkpoint(path)
    assert twin.global_step == 17 and twin.config['selected_properties'] == [0]
    assert twin.tokenizer.idx2piece == tokenizer.idx2piece and twin.tokenizer.pad_idx() == tokenizer.pad_idx()
    for (k, a), (_, b) in zip(model.state_dict().items(), twin.state_dict().items()):
        assert torch.equal(a, b), k
