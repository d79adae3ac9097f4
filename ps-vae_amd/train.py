"""Training entry: the fit / test loop the reference gets from pytorch_lightning.Trainer
(src/train.py:31-54), restated for a model that exposes the Lightning hooks
(training_step, validation_step, test_step, configure_optimizers, global_step, log):

  * per step: zero_grad, training_step, backward, global-norm clipping (gradient_clip_val,
    :47), optimiser step, global_step += 1;
  * per epoch: one pass over the validation loader (the batch-size weighted mean of the logged
    'val_loss' is the monitored value), the schedulers of configure_optimizers stepped once
    (LambdaLR 1 / (epoch + 1), src/pl_models/ps_vae_model.py:125-126), best-`monitor`
    checkpoint (ModelCheckpoint(monitor), :32-34) and VAE early stopping
    (src/utils/nn_utils.py:126-136: patience does not run while the KL weight is annealing);
  * afterwards one pass over the test loader (trainer.test, :54).

Several GPUs: one process per GPU under torchrun (the reference's own multi-GPU switch is
single-process DataParallel, :49-50; its one torch.distributed trainer is
src/guacamol_exps/goal_directed/train.py:159-161, 268-276, whose recipe this follows): the process
group is RCCL ('nccl'), the device comes from LOCAL_RANK, --batch_size is the global batch and
every rank takes batch_size / world molecules of its own share of the (identically shuffled)
training set, the gradients are averaged by one all-reduce inside the optimiser step, the
monitored value is reduced over the ranks' shares of the validation set, and rank 0 alone writes
checkpoints.

    python -m ps_vae_amd.train --train_set train.npz --valid_set valid.npz --test_set test.npz \
        --vocab vocab.txt --save_dir ckpt --alpha 0.1 [the flags of src/train.py:57-111]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m ps_vae_amd.train ...
"""
import argparse
import math
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def setup_seed(seed):
    """src/train.py:20-28."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def finish_anneal_step(args):
    """src/train.py:36, the same float expression (0.01 // 0.002 == 5.0: 4000 steps with the defaults)."""
    return args.kl_warmup + (args.max_beta // args.step_beta - 1) * args.kl_anneal_iter


class VAEEarlyStopping:
    """EarlyStopping(monitor, mode='min', min_delta=0, patience) whose best score and wait
    count are reset after every validation that happens before `finish_anneal_step`
    optimisation steps (src/utils/nn_utils.py:126-136)."""

    def __init__(self, finish_anneal_step, patience=3, min_delta=0.0):
        self.finish_anneal_step = finish_anneal_step
        self.patience, self.min_delta = patience, min_delta
        self.best_score, self.wait_count = math.inf, 0
        self.should_stop = False

    def on_validation_end(self, global_step, current):
        if not math.isfinite(current):          # Lightning's check_finite: stop on NaN / inf
            self.should_stop = True
        elif current < self.best_score - self.min_delta:
            self.best_score, self.wait_count = current, 0
        else:
            self.wait_count += 1
            if self.wait_count >= self.patience:
                self.should_stop = True
        if global_step < self.finish_anneal_step:     # a stop already requested above stands, as upstream
            self.best_score, self.wait_count = math.inf, 0
        return self.should_stop


class ModelCheckpoint:
    """Keeps the checkpoint with the lowest monitored value (save_top_k = 1): the model's
    state_dict under the reference's keys, epoch / global_step / monitor value, the optimiser
    state and - when the model offers hyper_parameters() - what the reference's
    save_hyperparameters() stores (config and vocabulary, src/pl_models/ps_vae_model.py:21), so
    that PSVAEModel.load_from_checkpoint can rebuild the model from the file alone.
    write=False (every rank but the first): the best score is tracked, nothing is written."""

    def __init__(self, dirpath, monitor='val_loss', write=True):
        self.dirpath, self.monitor, self.write = dirpath, monitor, write
        self.best_score, self.best_path = math.inf, None

    def on_validation_end(self, model, epoch, global_step, current, optimizer=None):
        if self.dirpath is None or not current < self.best_score:
            return None
        path = os.path.join(self.dirpath, f'epoch={epoch}-step={global_step}.ckpt')
        if self.write:
            os.makedirs(self.dirpath, exist_ok=True)
            payload = {'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
                       'epoch': epoch, 'global_step': global_step, self.monitor: current}
            if hasattr(model, 'hyper_parameters'):
                payload['hyper_parameters'] = model.hyper_parameters()
            if optimizer is not None:
                payload['optimizer'] = optimizer.state_dict()
            torch.save(payload, path)
            if self.best_path is not None and self.best_path != path and os.path.exists(self.best_path):
                os.remove(self.best_path)
        self.best_score, self.best_path = current, path
        return path


def _batch_size(batch):
    n = getattr(batch, 'batch_size', None)
    if n is None:
        n = len(batch['x']) if hasattr(batch, '__getitem__') else 1
    return int(n)


def _logged(model, name):
    v = model.logged[name]
    return float(v.detach()) if torch.is_tensor(v) else float(v)


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _reduce_device(model):
    """Where a tensor must live to be all-reduced: on this rank's GPU under RCCL, on the host under gloo."""
    if dist.get_backend() == 'nccl':
        return next(model.parameters()).device
    return torch.device('cpu')


@torch.no_grad()
def evaluate(model, loader, hook, name):
    """Batch-size weighted mean of the value `hook` logs under `name` over `loader`; with a
    process group, over the loaders of all ranks (sums of value x size and of sizes are
    all-reduced, so every rank returns the same number)."""
    was_training = model.training
    model.eval()
    total, count = 0.0, 0
    for i, batch in enumerate(loader):
        getattr(model, hook)(batch, i)
        n = _batch_size(batch)
        total += _logged(model, name) * n
        count += n
    model.train(was_training)
    if _world() > 1:
        t = torch.tensor([total, float(count)], dtype=torch.float64, device=_reduce_device(model))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total, count = float(t[0]), int(t[1])
    return total / count if count else math.nan


def fit(model, train_loader, valid_loader=None, test_loader=None, max_epochs=6, grad_clip=10.0, patience=3,
        finish_anneal=0, monitor='val_loss', save_dir=None, on_step=None, graph_steps=False):
    """Returns a dict: epochs run, global_step, per-epoch monitor values and learning rates,
    best checkpoint path, test loss.
    graph_steps (the fused optimiser, a loader over a BPEMolDataset): every full training batch is
    stepped by replaying the step recorded in a HIP graph (graph_step.GraphedTrainStep) - what small
    batches such as the reference's 32 need to keep the GPU busy.  With several ranks the recording
    covers forward, backward and gradient packing; the all-reduce and the optimiser follow it."""
    optimizers, schedulers = model.configure_optimizers()
    opt = optimizers[0]
    fused_tail = hasattr(opt, 'sync')          # optim.FlatAdam: exchange + clip + Adam in step()
    if fused_tail:
        opt.max_norm = grad_clip
    exchange = None
    if not fused_tail and _world() > 1:        # a stock optimiser: the same flat all-reduce, ahead of the clipping
        from .dist import FlatGradSync
        exchange = FlatGradSync(model.parameters())
    graphed = None
    if graph_steps:
        if not fused_tail or not hasattr(train_loader, 'order'):
            raise ValueError('graph_steps needs optim.FlatAdam and a loader of this package')
        from .graph_step import GraphedTrainStep
        graphed = GraphedTrainStep(model, opt, train_loader.dataset, train_loader.batch_size)
    stopper = VAEEarlyStopping(finish_anneal, patience=patience)
    ckpt = ModelCheckpoint(save_dir, monitor, write=_rank() == 0)
    history = {'monitor': [], 'lr': [], 'stopped_early': False}
    model.train()
    epoch = 0
    for epoch in range(max_epochs):
        history['lr'].append(opt.param_groups[0]['lr'])
        if graphed is not None:
            order = train_loader.order()
            train_loader.epoch += 1
            for i in range(0, len(order), train_loader.batch_size):
                loss = graphed.step(order[i:i + train_loader.batch_size])
                if on_step is not None:
                    on_step(model, loss)
        for i, batch in enumerate(train_loader if graphed is None else ()):
            if exchange is not None:
                exchange.zero_grad()
            else:
                opt.zero_grad()
            loss = model.training_step(batch, i)
            loss.backward()
            loss = loss.detach()   # the graph must be gone before the next forward builds its own
            if exchange is not None:
                exchange.sync()
            if not fused_tail and grad_clip:
                torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
            opt.step()
            model.global_step += 1
            if on_step is not None:
                on_step(model, loss)
        if valid_loader is not None:
            current = evaluate(model, valid_loader, 'validation_step', monitor)
            history['monitor'].append(current)
            ckpt.on_validation_end(model, epoch, model.global_step, current, opt)
            stopper.on_validation_end(model.global_step, current)
        for sched in schedulers:
            sched.step()
        if stopper.should_stop:
            history['stopped_early'] = True
            break
    history.update(epochs=epoch + 1, global_step=model.global_step, best_path=ckpt.best_path,
                   best_score=ckpt.best_score)
    if graphed is not None:
        history.update(graph_replays=graphed.replays, ordinary_steps=model.global_step - graphed.replays)
    if test_loader is not None:
        history['test_loss'] = evaluate(model, test_loader, 'test_step', 'test_loss')
    return history


def load_checkpoint(model, path, optimizer=None):
    """Restore a ModelCheckpoint file (tensors only: loaded with weights_only=True)."""
    payload = torch.load(path, map_location='cpu', weights_only=True)
    model.load_state_dict(payload['state_dict'])
    model.global_step = int(payload.get('global_step', 0))
    if optimizer is not None and 'optimizer' in payload:
        optimizer.load_state_dict(payload['optimizer'])
    return payload


def parse(argv=None):
    """The flags of src/train.py:57-111."""
    ap = argparse.ArgumentParser(description='training overall model for molecule generation')
    for name in ('train_set', 'valid_set', 'test_set', 'vocab', 'save_dir'):
        ap.add_argument('--' + name, type=str, required=True)
    ap.add_argument('--shuffle', action='store_true')
    ap.add_argument('--batch_size', type=int, default=32)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--alpha', type=float, required=True)
    ap.add_argument('--beta', type=float, default=0)
    ap.add_argument('--step_beta', type=float, default=0.002)
    ap.add_argument('--max_beta', type=float, default=0.01)
    ap.add_argument('--kl_warmup', type=int, default=0)
    ap.add_argument('--kl_anneal_iter', type=int, default=1000)
    ap.add_argument('--num_workers', type=int, default=4, help='accepted for compatibility: collation runs on the GPU')
    ap.add_argument('--gpus', default=None)
    ap.add_argument('--epochs', type=int, default=6)
    ap.add_argument('--patience', type=int, default=3)
    ap.add_argument('--grad_clip', type=float, default=10.0)
    ap.add_argument('--monitor', type=str, default='val_loss')
    ap.add_argument('--props', type=str, nargs='+', choices=['qed', 'sa', 'logp'], default=['qed', 'logp'])
    ap.add_argument('--predictor_hidden_dim', type=int, default=200)
    ap.add_argument('--node_hidden_dim', type=int, default=300)
    ap.add_argument('--graph_embedding_dim', type=int, default=400)
    ap.add_argument('--latent_dim', type=int, default=56)
    ap.add_argument('--max_pos', type=int, default=50)
    ap.add_argument('--atom_embedding_dim', type=int, default=50)
    ap.add_argument('--piece_embedding_dim', type=int, default=100)
    ap.add_argument('--pos_embedding_dim', type=int, default=50)
    ap.add_argument('--piece_hidden_dim', type=int, default=200)
    ap.add_argument('--no_graph_step', action='store_true',
                    help='launch the training step kernel by kernel even where it could be replayed from a HIP graph '
                         '(at most 256 molecules per rank and batch)')
    return ap.parse_args(argv)


def init_distributed(backend='nccl'):
    """Join the process group torchrun describes (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) before any
    other GPU call: the device is LOCAL_RANK and RCCL is bound to it.  Returns (rank, world, device);
    a plain `python -m ps_vae_amd.train` run is (0, 1, cuda:0 or the current device)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 1, torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    rank, local_rank = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    else:
        device = torch.device('cuda', local_rank) if torch.cuda.is_available() else torch.device('cpu')
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def main(argv=None):
    from .data import bpe_dataset
    from .pl_models import PSVAEModel
    from .utils.nn_utils import build_config
    from .utils.tokenizer import Tokenizer
    args = parse(argv)
    rank, world, device = init_distributed()
    setup_seed(2021)                         # the same initial weights on every rank
    tokenizer = Tokenizer(args.vocab)
    if args.batch_size % world:
        raise SystemExit(f'--batch_size {args.batch_size} is the global batch: it must divide by the {world} ranks')
    per_rank = args.batch_size // world      # src/guacamol_exps/goal_directed/train.py:272
    loaders = [bpe_dataset.get_dataloader(f, tokenizer, batch_size=per_rank, shuffle=sh, reference_layout=False,
                                          device=device, seed=2021, rank=rank, world=world, pad=pad)
               for f, sh, pad in ((args.train_set, args.shuffle, True), (args.valid_set, False, False),
                                  (args.test_set, False, False))]
    model = PSVAEModel(build_config(args, tokenizer), tokenizer).to(device)
    torch.manual_seed(2021 + rank)           # per-rank streams for the latent noise
    result = fit(model, *loaders, max_epochs=args.epochs, grad_clip=args.grad_clip, patience=args.patience,
                 finish_anneal=finish_anneal_step(args), monitor=args.monitor, save_dir=args.save_dir,
                 graph_steps=per_rank <= 256 and not args.no_graph_step)
    if rank == 0:
        print(result)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == '__main__':
    main()
