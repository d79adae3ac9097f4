"""The fit loop's host logic on CPU (ps_vae_amd.train): per-epoch LambdaLR, global-norm
clipping, best-val_loss checkpointing, VAE early stopping that waits for the KL annealing
(reference: src/train.py:31-54, src/utils/nn_utils.py:126-136,
src/pl_models/ps_vae_model.py:123-127), driven with a small stand-in model that exposes the
Lightning hooks."""
import math
import types

import pytest
import torch

import ps_vae_amd  # noqa: F401
from ps_vae_amd import train


class TinyModel(torch.nn.Module):
    """y = w x regression with the hook methods PSVAEModel has; validation losses are scripted."""

    def __init__(self, val_script):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([4.0]))
        self.global_step = 0
        self.logged = {}
        self.val_script = list(val_script)
        self.val_calls = 0
        self.grad_norms = []

    def log(self, name, value):
        self.logged[name] = value

    def training_step(self, batch, idx):
        loss = ((self.w * batch['x'] - batch['y']) ** 2).mean() * 1e4     # large: clipping is active
        self.log('loss', loss)
        return loss

    def validation_step(self, batch, idx):
        self.log('val_loss', torch.tensor(self.val_script[min(self.val_calls, len(self.val_script) - 1)], dtype=torch.float64))
        if idx == 1:
            self.val_calls += 1

    def test_step(self, batch, idx):
        self.log('test_loss', torch.tensor(0.5 + idx))

    def configure_optimizers(self):
        opt = torch.optim.Adam(self.parameters(), lr=1e-3)
        return [opt], [torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda epoch: 1 / (epoch + 1))]


def batches(n, size=4):
    return [{'x': torch.ones(size), 'y': torch.zeros(size)} for _ in range(n)]


def test_finish_anneal_step_matches_the_reference_expression():
    args = types.SimpleNamespace(kl_warmup=0, max_beta=0.01, step_beta=0.002, kl_anneal_iter=1000)
    assert train.finish_anneal_step(args) == 4000      # the reference's own expression: 0.01 // 0.002 == 5.0
    args.kl_warmup = 500
    assert train.finish_anneal_step(args) == 4500


def test_fit_schedules_clips_checkpoints_and_stops(tmp_path):
    # val_loss per epoch: improves twice, then stalls for `patience` epochs -> stop after epoch 4
    model = TinyModel([3.0, 2.0, 2.5, 2.2, 2.1, 1.0])
    seen = []

    def on_step(m, loss):
        seen.append(float(m.w.grad.abs()))

    out = train.fit(model, batches(5), batches(2), batches(2, size=3), max_epochs=10, grad_clip=10.0, patience=3,
                    finish_anneal=0, save_dir=str(tmp_path), on_step=on_step)
    assert out['stopped_early'] and out['epochs'] == 5 and out['global_step'] == 25
    assert out['monitor'] == [3.0, 2.0, 2.5, 2.2, 2.1]
    assert out['lr'] == pytest.approx([1e-3, 5e-4, 1e-3 / 3, 2.5e-4, 2e-4])          # LambdaLR 1 / (epoch + 1)
    assert all(g == pytest.approx(10.0, rel=1e-4) for g in seen)                       # clipped to norm 10
    assert out['best_score'] == 2.0 and out['best_path'].endswith('epoch=1-step=10.ckpt')
    assert [p.name for p in tmp_path.iterdir()] == ['epoch=1-step=10.ckpt']           # save_top_k = 1
    assert out['test_loss'] == pytest.approx(1.0)                                      # mean of 0.5, 1.5
    fresh = TinyModel([0.0])
    payload = train.load_checkpoint(fresh, out['best_path'])
    assert fresh.global_step == 10 and payload['val_loss'] == 2.0
    assert 'w' in payload['state_dict'] and 'optimizer' in payload


def test_early_stopping_waits_for_the_kl_annealing():
    # the same stalled curve, but annealing finishes at step 20 (= end of epoch 3): patience only
    # starts counting from the validation after that
    model = TinyModel([3.0, 2.0, 2.5, 2.2, 2.1, 2.3, 2.4, 2.6, 9.0])
    out = train.fit(model, batches(5), batches(2), max_epochs=12, patience=3, finish_anneal=20)
    # epochs 0-2 validate at steps 5, 10, 15 (< 20: reset every time); epoch 3 at step 20 sets the best
    # (2.2), epoch 4 improves (2.1), epochs 5, 6, 7 do not -> stop after epoch 7
    assert out['stopped_early'] and out['epochs'] == 8


def test_early_stopping_unit():
    es = train.VAEEarlyStopping(finish_anneal_step=100, patience=2)
    assert not es.on_validation_end(10, 1.0) and es.best_score == math.inf      # reset while annealing
    assert not es.on_validation_end(100, 1.0) and es.best_score == 1.0
    assert not es.on_validation_end(110, 1.5) and es.wait_count == 1
    assert es.on_validation_end(120, 1.0)                                       # not an improvement: min_delta 0
    assert train.VAEEarlyStopping(0, patience=5).on_validation_end(1, float('nan'))
