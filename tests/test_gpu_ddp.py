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
