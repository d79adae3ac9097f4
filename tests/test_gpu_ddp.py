"""Two ranks of the HIP model on ONE GPU, gradients exchanged through gloo (the staging branch of
FlatGradSync.sync): the data-parallel step as the 8-GPU job runs it, minus RCCL.  Each rank trains
on its own batch with the two-stream schedule on; after every step both ranks must hold
bit-identical parameters, the exchanged gradient must be the mean of the ranks' local gradients
(clipped to the global norm), and autograd must have had no stream complaint to make.

The ranks are started from a fork server that conftest launched before anything touched the GPU
(a process that has initialised the GPU must not exec another program on this pool)."""
import multiprocessing
import os
import socket
import warnings

import numpy as np
import pytest
import torch

WORLD, STEPS, BATCH, CLIP = 2, 2, 96, 10.0


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank(rank, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    import ps_vae_amd  # noqa: F401
    from ps_vae_amd.data.bpe_dataset import BPEMolDataset
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    from ps_vae_amd.pl_models import PSVAEModel
    from ps_vae_amd.utils.nn_utils import build_config, default_args
    from ps_vae_amd.utils.tokenizer import Tokenizer
    caught = []
    warnings.simplefilter('always')
    warnings.showwarning = lambda message, *a, **k: caught.append(str(message))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    dev = torch.device('cuda:0')
    tok = Tokenizer()
    torch.manual_seed(2021)                       # the same initial weights on every rank
    model = PSVAEModel(build_config(default_args(), tok), tok).to(dev)
    model.two_streams, model.two_streams_min_nodes = True, 0
    (opt,), _ = model.configure_optimizers()
    opt.max_norm = CLIP
    assert opt.sync.world == WORLD
    data = BPEMolDataset(None, tok, device=dev,
                         data=PackedMolecules.from_step1(synthetic_step1(BATCH, seed=100 + rank, shape='zinc')))
    torch.manual_seed(7 + rank)                   # rank-local latent noise
    out = {'local': [], 'exchanged': [], 'norm': [], 'params': []}
    for step in range(STEPS):
        batch = data.collate_ids(np.arange(BATCH))
        opt.zero_grad()
        model.training_step(batch, step).backward()
        opt.sync.pack()
        out['local'].append(opt.sync.flat.cpu())
        opt.step()
        model.global_step += 1
        out['exchanged'].append(opt.sync.flat.cpu())
        out['norm'].append(opt.grad_norm.cpu())
        out['params'].append(opt.sync.flat_params.cpu())
    torch.cuda.synchronize()
    out['warnings'] = [w for w in caught if 'AccumulateGrad' in w or 'stream' in w.lower()]
    torch.save(out, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_stay_identical(tmp_path):
    ctx = multiprocessing.get_context('forkserver')
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, port, str(tmp_path))) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, 'a rank did not finish'
    assert [p.exitcode for p in procs] == [0] * WORLD
    res = [torch.load(os.path.join(tmp_path, f'rank{r}.pt'), weights_only=True) for r in range(WORLD)]
    for step in range(STEPS):
        mean = (res[0]['local'][step].double() + res[1]['local'][step].double()) / WORLD
        norm = mean.norm()
        want = mean * min(1.0, CLIP / (norm.item() + 1e-6))
        assert not torch.equal(res[0]['local'][step], res[1]['local'][step])      # the ranks saw different data
        for r in range(WORLD):
            torch.testing.assert_close(res[r]['norm'][step].double(), norm, rtol=1e-5, atol=0)
            torch.testing.assert_close(res[r]['exchanged'][step].double(), want, rtol=1e-5, atol=1e-9)
        assert torch.equal(res[0]['exchanged'][step], res[1]['exchanged'][step])
        assert torch.equal(res[0]['params'][step], res[1]['params'][step])
    assert not torch.equal(res[0]['params'][0], res[0]['params'][1])              # and the step moved them
    assert res[0]['warnings'] == [] and res[1]['warnings'] == []


def _fit_rank(rank, port, out_dir, graph_steps=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      LOCAL_RANK='0')                  # both ranks drive GPU 0
    import torch.distributed as dist
    import ps_vae_amd  # noqa: F401
    from ps_vae_amd import train
    from ps_vae_amd.data.bpe_dataset import get_dataloader
    from ps_vae_amd.data.synthetic import synthetic_step1
    from ps_vae_amd.pl_models import PSVAEModel
    from ps_vae_amd.utils.nn_utils import build_config, default_args
    from ps_vae_amd.utils.tokenizer import Tokenizer
    got_rank, world, dev = train.init_distributed(backend='gloo')
    assert (got_rank, world, dev) == (rank, WORLD, torch.device('cuda', 0))
    tok = Tokenizer()
    train.setup_seed(2021)
    model = PSVAEModel(build_config(default_args(), tok), tok).to(dev)
    mols = synthetic_step1(101 + 37, seed=17, shape='zinc')
    seen = []
    loaders = [get_dataloader(None, tok, 16, shuffle=sh, device=dev, data=part, reference_layout=False, seed=3,
                              rank=rank, world=world, pad=pad)
               for part, sh, pad in ((mols[:101], True, True), (mols[101:], False, False))]
    order = loaders[0].order                            # record the ids of every epoch as the loader deals them
    loaders[0].order = lambda: seen.append(order().tolist()) or np.asarray(seen[-1])
    torch.manual_seed(2021 + rank)
    hist = train.fit(model, *loaders, max_epochs=2, grad_clip=10.0, save_dir=os.path.join(out_dir, 'ckpt'),
                     graph_steps=graph_steps)
    torch.cuda.synchronize()
    torch.save({'params': torch.cat([p.detach().flatten().cpu() for p in model.parameters()]), 'seen': seen,
                'monitor': hist['monitor'], 'steps': hist['global_step'], 'best_path': hist['best_path'],
                'replays': hist.get('graph_replays', 0)},
               os.path.join(out_dir, f'fit{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('graph_steps', [False, True])
def test_fit_two_ranks_on_one_gpu(tmp_path, graph_steps):
    """ps_vae_amd.train.fit with the HIP model as two ranks (gloo) on one GPU: disjoint shares of equal
    length per epoch, identical parameters after two epochs, one reduced validation value, one
    checkpoint (rank 0's) from which PSVAEModel.load_from_checkpoint rebuilds the model.
    graph_steps: full batches through the recorded step (forward, backward and packing replayed, the
    all-reduce and the optimiser behind it); the short last batch of an epoch the ordinary way."""
    ctx = multiprocessing.get_context('forkserver')
    port = _free_port()
    procs = [ctx.Process(target=_fit_rank, args=(r, port, str(tmp_path), graph_steps)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, 'a rank did not finish'
    assert [p.exitcode for p in procs] == [0] * WORLD
    res = [torch.load(os.path.join(tmp_path, f'fit{r}.pt'), weights_only=True) for r in range(WORLD)]
    assert torch.equal(res[0]['params'], res[1]['params'])
    assert res[0]['steps'] == res[1]['steps'] == 2 * 4                   # ceil(51 / 16) steps per epoch
    # three warm-up steps and the two short batches run the ordinary way, the other three are replays
    assert res[0]['replays'] == res[1]['replays'] == (3 if graph_steps else 0)
    assert res[0]['monitor'] == res[1]['monitor'] and np.isfinite(res[0]['monitor']).all()
    for e in range(2):
        a, b = res[0]['seen'][e], res[1]['seen'][e]
        assert len(a) == len(b) == 51 and sorted(set(a + b)) == list(range(101))
    assert res[0]['seen'][0] != res[0]['seen'][1]
    files = os.listdir(os.path.join(tmp_path, 'ckpt'))
    assert len(files) == 1 and res[0]['best_path'].endswith(files[0])
    from ps_vae_amd.pl_models import PSVAEModel
    twin = PSVAEModel.load_from_checkpoint(res[0]['best_path'])
    assert twin.global_step in (4, 8) and twin.tokenizer.num_piece_type() == 302
