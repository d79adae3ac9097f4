"""The batch-32 training step recorded once in a HIP graph over fixed-size padded buffers
(ps_vae_amd.graph_step.GraphedTrainStep) against the ordinary step and against the CPU oracle:
padding rows must take no part in any loss, gradient or update."""
import copy

import numpy as np
import pytest
import torch

from oracle import collate_ref, model_ref
from test_gpu_kernels import dev
from test_gpu_model import hip_model

pytestmark = pytest.mark.gpu


def _setup(tokenizer, shape, n_mol, seed):
    from ps_vae_amd.data.bpe_dataset import BPEMolDataset
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    step1 = synthetic_step1(n_mol, seed=seed, shape=shape)
    data = BPEMolDataset(None, tokenizer, device=dev(), data=PackedMolecules.from_step1(step1))
    return step1, data


@pytest.mark.parametrize('shape', ['zinc', 'qm9'])
def test_graphed_steps_follow_the_ordinary_steps(tokenizer, shape):
    """Eight steps at batch 32: the same molecules, uniforms and noise through the ordinary step and
    through the recorded one (3 ordinary warm-up steps on the padded buffers, then 5 replays).  The
    recorded model is LED: before every step it takes the ordinary model's weights and Adam moments,
    so each step is compared where both start from the same state - every loss within 1e-5, every
    clipped gradient within 1e-5 of its tensor's largest entry, the gradient norm within 1e-5, and
    the update within 1 % of lr wherever the gradient is clear of Adam's eps.  A batch that does not
    fit the fixed sizes falls back to the ordinary step."""
    from ps_vae_amd.graph_step import GraphedTrainStep
    B, steps = 32, 8
    _, data = _setup(tokenizer, shape, B * steps, seed=77)
    torch.manual_seed(3)
    state = hip_model(tokenizer)[0].state_dict()
    models = [hip_model(tokenizer, state=state)[0] for _ in range(2)]
    opts = []
    for m in models:
        (o,), _ = m.configure_optimizers()
        o.max_norm = 10.0
        m.config.update(kl_warmup=2, kl_anneal_iter=2, step_beta=0.002)      # the KL weight changes inside the run
        m.eps_tensor = torch.zeros(B, 56, device=dev())
        opts.append(o)
    graphed = GraphedTrainStep(models[1], opts[1], data, B)
    gen = torch.Generator().manual_seed(5)
    names = [k for k, p in models[0].named_parameters() if p.requires_grad]
    worst_g, worst_p = (0.0, ''), (0.0, '')
    for s in range(steps):
        ids = np.arange(s * B, (s + 1) * B)
        u = torch.rand(graphed.shapes.R + 8, generator=gen, dtype=torch.float64).to(dev())
        eps = torch.randn(B, 56, generator=gen).to(dev())
        for m in models:
            m.eps_tensor.copy_(eps)
        with torch.no_grad():                                  # lead the recorded model
            opts[1].sync.flat_params.copy_(opts[0].sync.flat_params)
            opts[1].exp_avg.copy_(opts[0].exp_avg)
            opts[1].exp_avg_sq.copy_(opts[0].exp_avg_sq)
        before = opts[0].sync.flat_params.clone()
        batch = data.collate_ids(ids, uniforms=u)
        opts[0].zero_grad()
        loss_a = models[0].training_step(batch, s)
        loss_a.backward()
        opts[0].step()
        models[0].global_step += 1
        loss_b = graphed.step(ids, uniforms=u)
        a, b = float(loss_a.detach()), float(loss_b)
        assert abs(a - b) <= 1e-5 * abs(a), (s, a, b)
        assert abs(float(opts[0].grad_norm) - float(opts[1].grad_norm)) <= 1e-5 * float(opts[0].grad_norm)
        for k, ga, gb, pa, pb, p0 in zip(names, opts[0].sync.views, opts[1].sync.views,
                                         (p for p in models[0].parameters() if p.requires_grad),
                                         (p for p in models[1].parameters() if p.requires_grad),
                                         (before[o:o + v.numel()].view_as(v) for o, v in
                                          zip(np.cumsum([0] + [(v.numel() + 3) // 4 * 4 for v in opts[0].sync.views[:-1]]),
                                              opts[0].sync.views))):
            scale = ga.abs().max().item()
            err = (ga - gb).abs().max().item()
            worst_g = max(worst_g, (err / max(scale, 1e-30), k))
            assert err <= 1e-5 * max(scale, 1e-30), (s, k, err, scale)
            clear = ga.abs() > 1e-5
            if bool(clear.any()):
                dp = (pa.detach() - pb.detach())[clear].abs().max().item()
                worst_p = max(worst_p, (dp / 1e-3, k))
                assert dp <= 0.01 * 1e-3, (s, k, dp)
                assert bool(((pa.detach() - p0)[clear] != 0).any())          # and the step did move them
    assert graphed.replays == steps - 3 and graphed.fallbacks == 0
    assert opts[0].steps == opts[1].steps == steps and models[1].global_step == steps
    print(f'{shape}: largest clipped-gradient difference {worst_g[0]:.2e} of its tensor\'s largest entry ({worst_g[1]}); '
          f'largest update difference {worst_p[0]:.2e} of lr ({worst_p[1]}); shapes {graphed.shapes}')
    # a batch beyond the fixed sizes: stepped the ordinary way
    small = copy.copy(graphed.shapes)
    small.V = 8
    tight = GraphedTrainStep(models[1], opts[1], data, B, shapes=small)
    tight.step(np.arange(B))
    assert tight.fallbacks == 1 and tight.replays == 0 and opts[1].steps == steps + 1


@pytest.mark.parametrize('shape', ['qm9', 'zinc'])
def test_graphed_step_against_oracle(tokenizer, shape):
    """One training step at batch 32 through the RECORDED path (after one ordinary warm-up step whose
    effect on weights and optimiser state is undone, the step is captured and replayed once)
    against the CPU oracle in float64 on the same molecules, uniforms and noise: every
    loss, the gradient norm, every clipped gradient and every parameter after the Adam step within
    1e-5 (relative to the tensor's largest entry)."""
    from ps_vae_amd.graph_step import GraphedTrainStep
    B = 32
    step1, data = _setup(tokenizer, shape, B, seed=55)
    uniforms = np.random.default_rng(8).random(40000)
    want_batch, used = collate_ref.collate(step1, uniforms, piece_pad=tokenizer.pad_idx())
    cfg = model_ref.default_config(tokenizer)
    torch.manual_seed(12)
    ref64 = model_ref.PSVAEModel(cfg, tokenizer).double()
    model, _ = hip_model(tokenizer, state={k: v.float() for k, v in ref64.state_dict().items()})
    ref64 = model_ref.PSVAEModel(cfg, tokenizer).double()
    ref64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
    (opt,), _ = model.configure_optimizers()
    opt.max_norm = 10.0
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(6))
    model.eps_tensor = eps.to(dev())
    graphed = GraphedTrainStep(model, opt, data, B, warmup=1)
    assert graphed.shapes.R <= uniforms.size
    u_dev = torch.from_numpy(uniforms).to(dev())
    start = {k: v.clone() for k, v in model.state_dict().items()}
    graphed.step(np.arange(B), uniforms=u_dev)                     # warm-up, then back to the start
    model.load_state_dict(start)
    opt.exp_avg.zero_(), opt.exp_avg_sq.zero_()
    opt.steps, model.global_step = 0, 0
    loss = graphed.step(np.arange(B), uniforms=u_dev)
    assert graphed.replays == 1 and data.collate_ids(np.arange(B), uniforms=torch.from_numpy(uniforms).to(dev())).num_uniforms == used
    tb = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in want_batch.items()}
    tb64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in tb.items()}
    pl, (rec, piece, edge, kl) = ref64(tb64, eps.double())
    want_loss = ref64.weighted_loss(pl, rec, kl)
    want_loss.backward()
    got = {'loss': float(loss), 'pred': float(model.logged['predictor loss']), 'rec': float(model.logged['reconstruct loss']),
           'piece': float(model.logged['piece loss']), 'edge': float(model.logged['edge loss']), 'kl': float(model.logged['kl loss'])}
    want = {'loss': want_loss.item(), 'pred': pl.item(), 'rec': rec.item(), 'piece': piece.item(), 'edge': edge.item(), 'kl': kl.item()}
    for k, v in want.items():
        assert abs(got[k] - v) <= 1e-5 * abs(v), (k, got[k], v)
    p64 = dict(ref64.named_parameters())
    g64 = {k: q.grad.clone() for k, q in p64.items() if q.grad is not None}
    norm64 = torch.nn.utils.clip_grad_norm_(ref64.parameters(), 10.0)
    assert abs(float(opt.grad_norm) - norm64.item()) <= 1e-5 * norm64.item()
    coef = min(1.0, 10.0 / (norm64.item() + 1e-6))
    worst = (0.0, '')
    for k, p in model.named_parameters():
        if k not in g64:
            assert not bool(p.grad.any()), k
            continue
        want_g = g64[k] * coef
        scale = want_g.abs().max().item()
        err = (p.grad.cpu().double() - want_g).abs().max().item()
        worst = max(worst, (err / max(scale, 1e-300), k))
        assert err <= 1e-5 * scale, f'clipped grad {k}: {err / scale:.2e}'
    torch.optim.Adam(ref64.parameters(), lr=cfg['lr']).step()
    for k, p in model.named_parameters():
        if k not in g64:
            continue
        clear = g64[k].abs() > 1e-5       # (where |g| is near Adam's eps the first step depends on the gradient's last bits)
        if not bool(clear.any()):
            continue
        want_p, got_p = p64[k].detach()[clear], p.detach().cpu().double()[clear]
        err = (got_p - want_p).abs().max().item()
        assert err <= 1e-5 * max(want_p.abs().max().item(), 1e-3), f'{k}: {err:.3e}'
    print(f'{shape}: recorded step, largest clipped-gradient error {worst[0]:.2e} of its tensor\'s largest entry ({worst[1]})')
