"""End-to-end parity on the GPU: the HIP-backed modules (reference API) against the golden
vectors the reference produced and against the CPU oracle on the same batches."""
import numpy as np
import pytest
import torch

from conftest import load_npz, sub
from oracle import collate_ref, model_ref
from test_gpu_kernels import close, dev

pytestmark = pytest.mark.gpu

# fp32 tolerance against the reference's own golden vectors (gradients, post-step parameters),
# relative to the tensor's largest entry
GOLD_REL = 1e-5

SMALL = dict(latent_dim=8, predictor_hidden_dim=16, node_hidden_dim=36, graph_embedding_dim=40,
             atom_embedding_dim=6, piece_embedding_dim=12, pos_embedding_dim=6, piece_hidden_dim=20)


def hip_model(tokenizer, state=None, seed=None, **dims):
    from ps_vae_amd.pl_models import PSVAEModel
    from ps_vae_amd.utils.nn_utils import build_config, default_args
    cfg = build_config(default_args(**dims), tokenizer)
    if seed is not None:
        torch.manual_seed(seed)
    model = PSVAEModel(cfg, tokenizer)
    if state is not None:
        res = model.load_state_dict(state, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    return model.to(dev()), cfg


def to_dev_batch(npb):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev()) for k, v in npb.items()}


def small_step1():
    from ps_vae_amd.data.synthetic import synthetic_step1
    return synthetic_step1(5, seed=31, shape='zinc') + synthetic_step1(3, seed=32, shape='qm9')


def test_state_dict_keys_match_reference(tokenizer):
    gold = load_npz('model_small.npz')
    model, _ = hip_model(tokenizer, **SMALL)
    assert set(model.state_dict().keys()) == set(sub(gold, 'state').keys())
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == gold[f'state/{k}'].shape, k


@pytest.mark.parametrize('path', ['reference_dict', 'compact'])
def test_small_model_against_reference_golden(tokenizer, path):
    gold = load_npz('model_small.npz')
    state = {k: torch.from_numpy(v) for k, v in sub(gold, 'state').items()}
    model, cfg = hip_model(tokenizer, state=state, **SMALL)
    eps = torch.from_numpy(gold['eps']).to(dev())
    if path == 'reference_dict':
        batch = to_dev_batch(sub(gold, 'batch'))
        with torch.no_grad():
            x = model.decoder.embed_atom(batch['x'], batch['x_pieces'], batch['x_pos'])
            assert torch.equal(x.cpu(), torch.from_numpy(gold['act/embed_atom']))
            B, N, D = x.shape
            last, all_x = model.encoder.embed_node(x.view(-1, D), batch['edge_index'], batch['edge_attr'])
            close(last, torch.from_numpy(gold['act/enc_last']), what='encoder last iterate')
            close(all_x, torch.from_numpy(gold['act/enc_all_x']), what='encoder all_x')
            gids = torch.repeat_interleave(torch.arange(B, device=dev()), N)
            genc = model.encoder.embed_graph(all_x, gids, batch['atom_mask'].flatten())
            close(genc, torch.from_numpy(gold['act/graph_embedding']), what='graph embedding')
            z, kl = model.decoder.rsample(genc, eps)
            close(z, torch.from_numpy(gold['act/z']), what='z')
            close(kl, torch.from_numpy(gold['act/kl']), what='kl')
            idx = batch['in_piece_edge_idx']
            h, _ = model.decoder.graph_embedding.embed_node(x.view(-1, D), batch['edge_index'][:, idx],
                                                            batch['edge_attr'][idx])
            close(h, torch.from_numpy(gold['act/dec_node_embedding']), what='decoder node embedding')
    else:
        from ps_vae_amd.data.collate import collate_packed
        from ps_vae_amd.data.packed import PackedMolecules
        packed = PackedMolecules.from_step1(small_step1())
        u = torch.from_numpy(gold['uniforms']).to(dev())
        batch = collate_packed(packed.to_device(dev()), np.arange(len(packed)), uniforms=u,
                               atom_pad=tokenizer.atom_pad_idx(), piece_pad=tokenizer.pad_idx(),
                               reference_layout=False)
        assert batch.compact is not None and 'x' not in batch

    flat_opt = None
    if path == 'compact':
        # the product's optimiser tail: flat parameters, clip + Adam in one kernel pair
        (flat_opt,), _ = model.configure_optimizers()
        flat_opt.max_norm = 10.0
        flat_opt.zero_grad()
    pred_loss, (losses, accus) = model(batch, return_accu=True, eps=eps)
    rec, piece, edge, kl = losses
    loss = model.weighted_loss(pred_loss, rec, kl)
    for name, val in [('total', loss), ('pred', pred_loss), ('rec', rec), ('piece', piece), ('edge', edge), ('kl', kl)]:
        np.testing.assert_allclose(val.item(), gold[f'loss/{name}'], rtol=1e-5, err_msg=name)
    assert accus[0] == pytest.approx(float(gold['accu/piece']))
    assert accus[1] == pytest.approx(float(gold['accu/edge']))
    loss.backward()
    no_grad = set(gold['no_grad_params'].tolist())
    for k, p in model.named_parameters():
        if k in no_grad:
            assert p.grad is None, k
            continue
        close(p.grad, torch.from_numpy(gold[f'grad/{k}']), rel=GOLD_REL, what=f'grad {k}')
    if flat_opt is not None:
        flat_opt.step()
        total = flat_opt.grad_norm
    else:
        opt = torch.optim.Adam(model.parameters(), lr=cfg['lr'])
        total = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
    np.testing.assert_allclose(total.item(), gold['grad_total_norm'], rtol=1e-5)
    for k, v in model.state_dict().items():
        close(v, torch.from_numpy(gold[f'after_step/{k}']), rel=GOLD_REL, what=f'after step {k}')


def test_flat_adam_follows_torch_adam_with_clipping():
    """Twenty steps of optim.FlatAdam (scale + clip + Adam on flat buffers) beside
    clip_grad_norm_ + torch.optim.Adam in float64 on the same gradients, clipping active on some
    steps and not on others, the learning rate changed on the way."""
    from ps_vae_amd.optim import FlatAdam
    d = dev()
    g = torch.Generator().manual_seed(0)
    shapes = [(300, 200), (300,), (7, 5), (1,), (1200, 13), (3,)]
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g, dtype=torch.float64)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().float().to(d)) for p in ref]
    ref_opt = torch.optim.Adam(ref, lr=1e-3)
    opt = FlatAdam(mine, lr=1e-3, max_norm=10.0)
    assert all(p.data_ptr() % 16 == 0 for p in mine)
    for step in range(20):
        scale = 100.0 if step % 3 == 0 else 0.01          # norm above / below the clipping threshold
        grads = [torch.randn(*s, generator=g, dtype=torch.float64) * scale for s in shapes]
        if step == 10:
            for o in (ref_opt, opt):
                o.param_groups[0]['lr'] = 5e-4
        opt.zero_grad()
        for p, q, gr in zip(ref, mine, grads):
            p.grad = gr.clone()
            q.grad = gr.float().to(d)
        if step % 5 == 4:
            ref[3].grad = None
            mine[3].grad = None                            # a parameter without a gradient this step
            # (torch skips it; the flat update sees a zero gradient: the moments decay, which
            # moves the parameter by its running mean - compare everything else)
        want_norm = torch.nn.utils.clip_grad_norm_([p for p in ref if p.grad is not None], 10.0)
        ref_opt.step()
        opt.step()
        np.testing.assert_allclose(opt.grad_norm.item(), want_norm.item(), rtol=2e-6)
        for i, (p, q) in enumerate(zip(ref, mine)):
            if i == 3:
                continue
            close(q, p, rel=2e-6, what=f'step {step} param {i}')
            if p.grad is not None:
                close(q.grad, p.grad, rel=2e-6, what=f'step {step} clipped grad {i}')
    # the optimiser state travels in torch.optim.Adam's own layout, both ways
    state = opt.state_dict()
    want = ref_opt.state_dict()
    assert state['param_groups'][0]['params'] == want['param_groups'][0]['params'] == list(range(len(shapes)))
    assert state['param_groups'][0]['lr'] == want['param_groups'][0]['lr'] == 5e-4
    for i in range(len(shapes)):
        if i == 3:
            continue
        assert float(state['state'][i]['step']) == float(want['state'][i]['step']) == 20
        close(state['state'][i]['exp_avg'], want['state'][i]['exp_avg'], rel=2e-6, what=f'exp_avg {i}')
        close(state['state'][i]['exp_avg_sq'], want['state'][i]['exp_avg_sq'], rel=2e-6, what=f'exp_avg_sq {i}')
    stock = torch.optim.Adam([torch.nn.Parameter(q.detach().clone()) for q in mine], lr=1e-3)
    stock.load_state_dict(state)                               # torch's Adam reads it
    twin = FlatAdam([torch.nn.Parameter(q.detach().clone()) for q in mine], lr=1e-3, max_norm=10.0)
    twin.load_state_dict(want)                                 # and a state written by torch's Adam is read here
    assert twin.steps == 20 and twin.param_groups[0]['lr'] == 5e-4
    mask = torch.ones_like(opt.exp_avg, dtype=torch.bool)
    off = sum((torch.Size(s).numel() + 3) // 4 * 4 for s in shapes[:3])
    mask[off:off + 4] = False                                  # (parameter 3: torch skipped its gradient-less steps)
    close(twin.exp_avg[mask], opt.exp_avg[mask], rel=2e-6, what='moments read back from torch')
    twin.load_state_dict(state)
    assert torch.equal(twin.exp_avg, opt.exp_avg) and torch.equal(twin.exp_avg_sq, opt.exp_avg_sq) and twin.steps == opt.steps


def test_default_width_model_against_reference_golden(tokenizer):
    """H=300, G=400, L=56 — the awkward widths the kernels are built for."""
    gold = load_npz('model_default.npz')
    model, _ = hip_model(tokenizer, seed=int(gold['seed']))
    batch = to_dev_batch(sub(gold, 'batch'))
    eps = torch.from_numpy(gold['eps']).to(dev())
    pred_loss, (rec, piece, edge, kl) = model(batch, eps=eps)
    loss = model.weighted_loss(pred_loss, rec, kl)
    for name, val in [('total', loss), ('pred', pred_loss), ('rec', rec), ('piece', piece), ('edge', edge), ('kl', kl)]:
        np.testing.assert_allclose(val.item(), gold[f'loss/{name}'], rtol=1e-5, err_msg=name)
    loss.backward()
    worst = (0.0, '')
    for k, p in model.named_parameters():
        if p.grad is None:
            assert k in set(gold['no_grad_params'].tolist())
            continue
        np.testing.assert_allclose(p.grad.norm().item(), gold[f'gradnorm/{k}'], rtol=1e-4, err_msg=k)
        # element-wise against the reference itself: every 97th entry of every gradient at the
        # widths the kernels are tuned for, within GOLD_REL of the tensor's largest entry
        got = p.grad.reshape(-1)[::97].cpu().numpy()
        err = np.abs(got - gold[f'gradsample/{k}']).max()
        worst = max(worst, (err / float(gold[f'gradmax/{k}']), k))
        assert err <= GOLD_REL * float(gold[f'gradmax/{k}']), f'{k}: {err:.3e} vs max {float(gold[f"gradmax/{k}"]):.3e}'
    (opt,), _ = model.configure_optimizers()          # also packs the gradients just produced
    opt.max_norm = 10.0
    opt.step()
    np.testing.assert_allclose(opt.grad_norm.item(), gold['grad_total_norm'], rtol=1e-5)
    # Adam's first step moves an entry by lr g / (|g| + eps): where |g| is within a few orders of eps = 1e-8 the
    # step amplifies a gradient error dg by lr eps / (|g| + eps)^2.  The bound of an entry is GOLD_REL of the
    # tensor's largest entry plus that amplification of the gradient error allowed above (never more than the
    # 2 lr a step can differ by); the count of entries that needed the second term is printed.
    lr, adam_eps, widened = float(opt.param_groups[0]['lr']), 1e-8, 0
    for k, v in model.state_dict().items():
        want = gold[f'after_step_sample/{k}']
        got = v.reshape(-1)[::97].cpu().numpy()
        err = np.abs(got - want)
        base = GOLD_REL * max(np.abs(want).max(), 1e-30)
        bound = np.full_like(err, base)
        if f'gradsample/{k}' in gold:
            g = np.abs(gold[f'gradsample/{k}'])
            bound = base + np.minimum(2 * lr, lr * adam_eps * GOLD_REL * float(gold[f'gradmax/{k}']) / (g + adam_eps) ** 2)
        assert (err <= bound).all(), f'after step {k}: {err.max():.3e} vs {base:.3e}'
        widened += int((err > base).sum())
    print(f'default widths vs the reference: largest sampled gradient error {worst[0]:.2e} of its tensor\'s largest entry ({worst[1]}); '
          f'{widened} sampled entries after the step lie within the bound only with Adam\'s amplification of a near-zero gradient')


def test_default_width_batch_256_against_oracle(tokenizer):
    """A larger batch through the compact path vs the CPU oracle on the reference layout of
    the same collation: every loss and every parameter gradient.

    Yardstick: the oracle evaluated in float64 (the fp32 CPU oracle itself is off the float64
    result by up to ~3e-3 of the tensor's largest entry on decoder.piece_embedding.weight, so it
    cannot be the yardstick): every gradient must be within 1e-5 of the float64 value, relative to
    the tensor's largest entry.  The fp32 CPU oracle's own distance is printed beside it.

    ReLU kinks: with ~16 M ReLU inputs in this batch a few lie within fp32 rounding of zero,
    and an implementation that rounds differently takes the other side of the kink (the loss
    is not differentiable there).  The test counts the ReLU outputs whose sign differs from
    the float64 oracle's (at most 8) and compares the gradients with the float64 oracle
    re-evaluated under the GPU's own ReLU decisions, so the bound is NOT relaxed for a flip."""
    import copy
    from ps_vae_amd import ops as hip_ops
    from ps_vae_amd.data.collate import collate_packed
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    B = 256
    step1 = synthetic_step1(B, seed=123, shape='zinc')
    uniforms = np.random.default_rng(1).random(700 * B)
    want_batch, used = collate_ref.collate(step1, uniforms, piece_pad=tokenizer.pad_idx())
    cfg = model_ref.default_config(tokenizer)
    torch.manual_seed(7)
    ref = model_ref.PSVAEModel(cfg, tokenizer)
    ref64 = copy.deepcopy(ref).double()
    model, _ = hip_model(tokenizer, state=ref.state_dict())
    model.two_streams_min_nodes = 0          # exercise the two-stream schedule at this batch size too
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(2))
    tb = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in want_batch.items()}
    pl, (rec, piece, edge, kl) = ref(tb, eps)
    ref_loss = ref.weighted_loss(pl, rec, kl)
    ref_loss.backward()

    relu64 = []      # float64 ReLU outputs in call order, real atoms / pairs only
    real = ~tb['atom_mask'].flatten()

    def rec_relu(mod, inp, out):
        relu64.append(out.detach()[real] if out.shape[0] == real.numel() else out.detach())
    hooks = [m.register_forward_hook(rec_relu) for m in ref64.modules() if isinstance(m, torch.nn.ReLU)]
    pl64, (rec64, piece64, edge64, kl64) = ref64(tb, eps.double())
    for h in hooks:
        h.remove()
    ref64.weighted_loss(pl64, rec64, kl64).backward()

    packed = PackedMolecules.from_step1(step1)
    batch = collate_packed(packed.to_device(dev()), np.arange(B), uniforms=torch.from_numpy(uniforms).to(dev()),
                           atom_pad=tokenizer.atom_pad_idx(), piece_pad=tokenizer.pad_idx(), reference_layout=False)
    assert batch.num_uniforms == used
    hip_ops.RELU_TRACE = []
    try:
        gpl, (grec, gpiece, gedge, gkl) = model(batch, eps=eps.to(dev()))
        relu_gpu = list(hip_ops.RELU_TRACE)
    finally:
        hip_ops.RELU_TRACE = None
    loss = model.weighted_loss(gpl, grec, gkl)
    loss.backward()
    for name, a, b in [('total', loss, ref_loss), ('pred', gpl, pl), ('piece', gpiece, piece), ('edge', gedge, edge),
                       ('kl', gkl, kl)]:
        np.testing.assert_allclose(a.item(), b.item(), rtol=1e-5, err_msg=name)

    # the oracle calls its ReLU modules in the same order as the HIP model runs its fused
    # Linear+ReLU layers: encoder x4, decoder GNN x4, link MLP x2, predictor x2
    # (with the decoder GNN forked to its side stream it is launched before the encoder)
    if model.two_streams:
        relu_gpu = relu_gpu[4:8] + relu_gpu[0:4] + relu_gpu[8:]
    assert len(relu_gpu) == len(relu64)
    flips = 0
    for a, b in zip(relu_gpu, relu64):
        assert a.shape == b.shape
        flips += ((a.cpu() > 0) != (b > 0)).sum().item()
    assert flips <= 8, f'{flips} ReLU outputs on the other side of the kink'

    # The yardstick for the gradients: the float64 oracle evaluated with the GPU's own ReLU
    # decisions forced (every ReLU output = input x [GPU output > 0]).  Where the GPU took the
    # same side of every kink this IS the float64 oracle; where a pre-activation within fp32
    # rounding of zero fell on the other side, it is the derivative of the function the GPU
    # actually evaluated - so a gradient error that is NOT explained by such a flip still fails.
    ref64f = copy.deepcopy(ref).double()
    for q in ref64f.parameters():
        q.grad = None
    masks = iter([(a > 0).cpu() for a in relu_gpu])

    def force(mod, inp, out):
        x, m = inp[0], next(masks)
        if x.shape[0] == real.numel():
            full = x > 0
            full[real] = m
        else:
            full = m
        return x * full
    hooks = [m.register_forward_hook(force) for m in ref64f.modules() if isinstance(m, torch.nn.ReLU)]
    plf, (recf, _, _, klf) = ref64f(tb, eps.double())
    for h in hooks:
        h.remove()
    ref64f.weighted_loss(plf, recf, klf).backward()

    refp, refp64, refp64f = dict(ref.named_parameters()), dict(ref64.named_parameters()), dict(ref64f.named_parameters())
    worst, worst_cpu = (0.0, ''), (0.0, '')
    for k, p in model.named_parameters():
        if refp[k].grad is None:
            assert p.grad is None, k
            continue
        g64, g64f = refp64[k].grad, refp64f[k].grad
        scale = g64.abs().max().item()
        err_gpu = (p.grad.cpu().double() - g64f).abs().max().item()
        err_cpu = (refp[k].grad.double() - g64).abs().max().item()
        worst = max(worst, (err_gpu / scale, k))
        worst_cpu = max(worst_cpu, (err_cpu / scale, k))
        assert err_gpu <= 1e-5 * scale, \
            f'grad {k}: GPU {err_gpu / scale:.2e} of the largest entry from the float64 oracle under the GPU\'s ' \
            f'ReLU decisions; fp32 CPU oracle {err_cpu / scale:.2e} from the plain float64 oracle; {flips} ReLU flips'
    print(f'{flips} ReLU flips; largest GPU gradient error {worst[0]:.2e} of its tensor\'s largest entry ({worst[1]}); '
          f'the fp32 CPU oracle\'s own largest distance from float64: {worst_cpu[0]:.2e} ({worst_cpu[1]})')


def test_loss_curve_follows_the_oracle_over_30_steps(tokenizer):
    """Thirty optimisation steps (batch 32, default widths, KL weight stepping on from step 5)
    on the GPU path - GPU collation, flat-gradient clip, fused Adam - beside the CPU oracle
    trained in float64 on the same molecules, uniforms and eps.

    Two GPU models run the schedule.
    * The LED one starts every step from the float64 oracle's weights and Adam moments
      (rounded to fp32), so each of the thirty steps is compared where both sides still hold
      the same weights: every loss and the clipped gradient norm within 1e-5 relative of
      float64, and the update itself against Adam evaluated in float64 on the model's own
      clipped gradient (weights within 2e-6 lr, both moments within 1e-6).
    * The FREE one trains on its own.  Its curve can only be held loosely, because fp32
      trajectories of this model separate in jumps: Adam's first steps move every weight by
      about lr whatever the size of its gradient, and the decoder's message-passing weights
      have gradient entries of the size of fp32 rounding, so rounding decides which way some
      weights go.  The fp32 CPU oracle itself shows it - from initial weights one rounding step
      (relative 2^-23) away from the given ones, two of five runs are 4e-5 / 6e-5 off the
      float64 edge loss at step 3 and 1e-4 at step 5 while the others are still at 1e-6, and
      which runs jump changes with the host's BLAS.  So the free model is held to 1e-5 at step
      0 (same weights); afterwards the total, reconstruction, piece and edge losses to 1 % of the
      float64 curve (observed: up to 3e-4) and the two small terms that amplify the drift most,
      the property loss and the KL term (the fp32 CPU oracle's KL is 1e-2 off by step 25, the
      GPU's up to 5e-2), to 20 %; and the curve must fall."""
    import copy
    from ps_vae_amd.data.collate import collate_packed
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    steps, B = 30, 32
    sched = dict(kl_warmup=5, kl_anneal_iter=5, step_beta=0.002, max_beta=0.01)
    cfg = model_ref.default_config(tokenizer, **sched)
    torch.manual_seed(3)
    ref = model_ref.PSVAEModel(cfg, tokenizer)
    ref64 = copy.deepcopy(ref).double()
    ref_opt = torch.optim.Adam(ref.parameters(), lr=cfg['lr'])
    ref64_opt = torch.optim.Adam(ref64.parameters(), lr=cfg['lr'])
    led, _ = hip_model(tokenizer, state=ref.state_dict(), **sched)
    free, _ = hip_model(tokenizer, state=ref.state_dict(), **sched)
    (led_opt,), _ = led.configure_optimizers()      # optim.FlatAdam: exchange + clip + Adam on flat buffers
    (free_opt,), _ = free.configure_optimizers()
    led_opt.max_norm = free_opt.max_norm = 10.0
    p64 = dict(ref64.named_parameters())
    led_named = [(k, q) for k, q in led.named_parameters() if q.requires_grad]
    assert [id(q) for _, q in led_named] == [id(q) for q in led_opt.sync.params]
    offs, total = [], 0
    for _, q in led_named:
        offs.append(total)
        total += (q.numel() + 3) // 4 * 4

    def lead():
        """weights, Adam moments and step count of the float64 oracle -> the led GPU model"""
        with torch.no_grad():
            for (k, q), o in zip(led_named, offs):
                q.copy_(p64[k].detach().to(torch.float32))
                st = ref64_opt.state.get(p64[k], {})
                for name, flat in (('exp_avg', led_opt.exp_avg), ('exp_avg_sq', led_opt.exp_avg_sq)):
                    slot = flat[o:o + q.numel()].view_as(q)
                    if name in st:
                        slot.copy_(st[name].to(torch.float32))
                    else:
                        slot.zero_()
        led_opt.steps = led.global_step = ref64.global_step

    mols = synthetic_step1(steps * B, seed=77, shape='zinc')
    rng = np.random.default_rng(9)
    gen = torch.Generator().manual_seed(4)
    names = ['loss', 'pred', 'rec', 'piece', 'edge', 'kl']
    report, curve = [], []
    worst_led, worst_update = (0.0, ''), (0.0, '')
    for s in range(steps):
        chunk = mols[s * B:(s + 1) * B]
        uniforms = rng.random(700 * B)
        eps = torch.randn(B, 56, generator=gen)
        ref_batch, used = collate_ref.collate(chunk, uniforms, piece_pad=tokenizer.pad_idx())
        tb = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ref_batch.items()}
        tb64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in tb.items()}
        packed = PackedMolecules.from_step1(chunk)
        batch = collate_packed(packed.to_device(dev()), np.arange(B), uniforms=torch.from_numpy(uniforms).to(dev()),
                               atom_pad=tokenizer.atom_pad_idx(), piece_pad=tokenizer.pad_idx(),
                               reference_layout=False)
        assert batch.num_uniforms == used
        lead()
        before = {'p': led_opt.sync.flat_params.double(), 'm': led_opt.exp_avg.double(), 'v': led_opt.exp_avg_sq.double()}
        beta_ref = ref64.cal_beta()
        cpu32 = model_ref.train_step(ref, ref_opt, tb, eps)
        # the float64 step, spelt out to keep the gradient norm and the moments it leaves
        ref64_opt.zero_grad(set_to_none=True)
        pl64, (rec64, piece64, edge64, kl64) = ref64(tb64, eps.double())
        loss64 = ref64.weighted_loss(pl64, rec64, kl64)
        loss64.backward()
        norm64 = torch.nn.utils.clip_grad_norm_(ref64.parameters(), 10.0).item()
        ref64_opt.step()
        ref64.global_step += 1
        cpu64 = dict(zip(names, (v.item() for v in (loss64, pl64, rec64, piece64, edge64, kl64))))

        def gpu_step(model, opt):
            opt.zero_grad()
            pred_loss, (rec, piece, edge, kl) = model(batch, eps=eps.to(dev()))
            assert abs(model.cal_beta() - beta_ref) < 1e-12
            loss = model.weighted_loss(pred_loss, rec, kl)
            loss.backward()
            opt.step()
            model.global_step += 1
            return dict(zip(names, (v.item() for v in (loss, pred_loss, rec, piece, edge, kl))))
        got_led, got_free = gpu_step(led, led_opt), gpu_step(free, free_opt)
        for k in names:
            scale = max(abs(cpu64[k]), 1e-12)
            d_led = abs(got_led[k] - cpu64[k]) / scale
            worst_led = max(worst_led, (d_led, f'step {s} {k}'))
            assert d_led <= 1e-5, f'led model, step {s} {k}: GPU {got_led[k]:.7f}, float64 oracle {cpu64[k]:.7f}'
            d_free, d_cpu = abs(got_free[k] - cpu64[k]) / scale, abs(cpu32[k] - cpu64[k]) / scale
            report.append((s, k, d_free, d_cpu))
            assert d_free <= (1e-5 if s == 0 else (1e-2 if k in ('loss', 'rec', 'piece', 'edge') else 2e-1)), \
                f'free model, step {s} {k}: GPU {got_free[k]:.6f}, fp32 oracle {cpu32[k]:.6f}, float64 oracle {cpu64[k]:.6f}'
        assert abs(float(led_opt.grad_norm) - norm64) <= 1e-5 * norm64, (s, float(led_opt.grad_norm), norm64)
        # the update itself: Adam in float64 from the led model's OWN clipped gradient (left in the
        # flat buffer) and the moments / weights it started the step with
        t, lr = ref64.global_step, cfg['lr']
        g = led_opt.sync.flat.double()
        m = 0.9 * before['m'] + 0.1 * g
        v = 0.999 * before['v'] + 0.001 * g * g
        want_p = before['p'] - (lr / (1 - 0.9 ** t)) * m / ((v / (1 - 0.999 ** t)).sqrt() + 1e-8)
        err = ((led_opt.sync.flat_params.double() - want_p).abs() - 1e-7 * want_p.abs()).max().item()
        worst_update = max(worst_update, (err / lr, f'step {s}'))
        assert err <= 2e-6 * lr, f'led model, weights after step {s}: {err / lr:.2e} of lr off Adam in float64'
        m_size = 0.9 * before['m'].abs() + 0.1 * g.abs()      # (the two terms of m may cancel; those of v cannot)
        assert bool(((led_opt.exp_avg.double() - m).abs() <= 1e-6 * m_size + 1e-30).all()), f'first moment after step {s}'
        assert bool(((led_opt.exp_avg_sq.double() - v).abs() <= 1e-6 * v + 1e-30).all()), f'second moment after step {s}'
        curve.append(cpu64['loss'])
    assert curve[-1] < 0.9 * curve[0], (curve[0], curve[-1])      # the curve actually moves over these steps
    print(f'led model: largest loss distance from float64 {worst_led[0]:.2e} ({worst_led[1]}), '
          f'largest distance of an update from Adam in float64 {worst_update[0]:.2e} lr ({worst_update[1]})')
    for st in list(range(0, 5)) + list(range(5, steps, 5)):     # free model / fp32 CPU oracle: distance from the float64 curve
        rows = [r for r in report if r[0] == st]
        print(st, ' '.join(f'{r[1]}:{r[2]:.1e}/{r[3]:.1e}' for r in rows))
    worst = max(report, key=lambda r: r[2])
    print(f'free model: largest deviation from the float64 curve: step {worst[0]} {worst[1]} {worst[2]:.2e} '
          f'(fp32 CPU oracle there: {worst[3]:.2e}); loss {curve[0]:.4f} -> {curve[-1]:.4f}')


@pytest.mark.parametrize('shape', ['qm9', 'zinc'])
def test_batch_32_train_step_against_oracle(tokenizer, shape):
    """One full training step at batch 32, default widths, on QM9-shaped molecules (at most 9
    atoms: the reference's own CPU configuration) and ZINC-shaped ones: GPU collation bit-exact
    against the oracle's, every loss, every parameter gradient, the clipped gradient norm and
    every parameter after the Adam step against the oracle in float64."""
    from ps_vae_amd.data.synthetic import synthetic_step1
    batch_32_step_against_oracle(tokenizer, synthetic_step1(32, seed=55, shape=shape))


@pytest.mark.parametrize('batch', [32, 512, 4096])
def test_training_repeats_bit_for_bit(tokenizer, batch):
    """Two runs of eight training steps from the same seed on the same batches end in the same parameters, bit
    for bit, with the two-stream fork on: no kernel of the step adds in an order that depends on timing (no
    atomics: table gradients, pooled sums and split-K slabs are summed in fixed order), and a missing stream
    dependency between the two branches would show up here as a difference in some run."""
    from ps_vae_amd.data.bpe_dataset import BPEMolDataset
    from ps_vae_amd.data.packed import PackedMolecules
    from ps_vae_amd.data.synthetic import synthetic_step1
    ds = BPEMolDataset(None, tokenizer, device=dev(), data=PackedMolecules.from_step1(synthetic_step1(2 * batch, seed=77, shape='zinc')))
    ends = []
    for run in range(3):
        model, _ = hip_model(tokenizer, seed=4)
        model.two_streams_min_nodes = 0
        (opt,), _ = model.configure_optimizers()
        opt.max_norm = 10.0
        model.train()
        rng = np.random.default_rng(5)
        torch.manual_seed(6)                      # the noise of the reparameterisation
        for step in range(8):
            ids = rng.choice(len(ds), size=batch, replace=False)
            uniforms = torch.rand(600 * batch, dtype=torch.float64, device=dev(), generator=torch.Generator(device=dev()).manual_seed(step))
            opt.zero_grad()
            loss = model.training_step(ds.collate_ids(ids, reference_layout=False, uniforms=uniforms), step)
            loss.backward()
            opt.step()
            model.global_step += 1
        torch.cuda.synchronize()
        ends.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    for other in ends[1:]:
        bad = [k for k in ends[0] if not torch.equal(ends[0][k], other[k])]
        assert not bad, f'parameters differ between two identical runs: {bad[:5]}'


@pytest.mark.parametrize('name', ['qm9', 'zinc'])
def test_batch_32_train_step_on_real_molecules(name):
    """The same step on REAL molecules: the first 32 SMILES of the reference's data/qm9/train.txt and
    data/zinc250k/valid.txt (tests/golden/real/), read, kekulised and tokenised without RDKit
    (data.mol_bpe) against a 300-piece vocabulary mined from the first 2048 lines of the same file.
    The preprocessing is not pinned by the reference (no RDKit, no vocabulary file ships with it); the
    kernels are held to the oracle on whatever molecules it yields.  Property scores are zeros."""
    import os
    from ps_vae_amd.data import mol_bpe
    from ps_vae_amd.utils.tokenizer import Tokenizer
    real = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'real')
    tok = Tokenizer(os.path.join(real, f'{name}_vocab_300.txt'))
    with open(os.path.join(real, f'{name}_2048.smi')) as f:
        smiles = [line.strip() for line in f][:32]
    np.random.seed(4)                       # the order of a molecule's pieces is shuffled (mol_bpe.py:214)
    batch_32_step_against_oracle(tok, [mol_bpe.process_step1(s, tok) for s in smiles])


def batch_32_step_against_oracle(tokenizer, step1):
    import copy
    from ps_vae_amd.data.collate import collate_packed
    from ps_vae_amd.data.packed import PackedMolecules
    B = len(step1)
    uniforms = np.random.default_rng(8).random(1500 * B)
    want_batch, used = collate_ref.collate(step1, uniforms, piece_pad=tokenizer.pad_idx())
    cfg = model_ref.default_config(tokenizer)
    torch.manual_seed(12)
    ref = model_ref.PSVAEModel(cfg, tokenizer)
    ref64 = copy.deepcopy(ref).double()
    model, _ = hip_model(tokenizer, state=ref.state_dict())
    (opt,), _ = model.configure_optimizers()
    opt.max_norm = 10.0
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(6))
    tb = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in want_batch.items()}
    tb64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in tb.items()}

    packed = PackedMolecules.from_step1(step1)
    batch = collate_packed(packed.to_device(dev()), np.arange(B), uniforms=torch.from_numpy(uniforms).to(dev()),
                           atom_pad=tokenizer.atom_pad_idx(), piece_pad=tokenizer.pad_idx())
    assert batch.num_uniforms == used
    for k, v in want_batch.items():       # the reference layout, bit for bit
        got = batch[k].cpu().numpy()
        assert got.shape == v.shape and np.array_equal(got, v), k

    def run(m, b, e):
        pl, (rec, piece, edge, kl) = m(b, e)
        loss = m.weighted_loss(pl, rec, kl)
        loss.backward()
        return dict(loss=loss.item(), pred=pl.item(), rec=rec.item(), piece=piece.item(), edge=edge.item(), kl=kl.item())
    l32, l64 = run(ref, tb, eps), run(ref64, tb64, eps.double())
    opt.zero_grad()
    gpl, (grec, gpiece, gedge, gkl) = model(batch, eps=eps.to(dev()))
    gloss = model.weighted_loss(gpl, grec, gkl)
    gloss.backward()
    got = dict(loss=gloss.item(), pred=gpl.item(), rec=grec.item(), piece=gpiece.item(), edge=gedge.item(), kl=gkl.item())
    for k, v in l64.items():
        assert abs(got[k] - v) <= 1e-5 * abs(v), (k, got[k], v, l32[k])
    p32, p64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
    worst = (0.0, '', 0.0)
    for k, p in model.named_parameters():
        if p64[k].grad is None:
            assert p.grad is None or not bool(p.grad.any()), k
            continue
        g64 = p64[k].grad
        scale = g64.abs().max().item()
        err_gpu = (p.grad.cpu().double() - g64).abs().max().item()
        err_cpu = (p32[k].grad.double() - g64).abs().max().item()
        worst = max(worst, (err_gpu / scale, k, err_cpu / scale))
        assert err_gpu <= 1e-5 * scale, f'grad {k}: {err_gpu / scale:.2e} (fp32 oracle {err_cpu / scale:.2e})'
    print(f'largest GPU gradient error {worst[0]:.2e} of its tensor\'s largest entry ({worst[1]}; fp32 CPU oracle there {worst[2]:.2e})')
    g64 = {k: q.grad.clone() for k, q in p64.items() if q.grad is not None}
    norm64 = torch.nn.utils.clip_grad_norm_(ref64.parameters(), 10.0)
    torch.optim.Adam(ref64.parameters(), lr=cfg['lr']).step()
    opt.step()
    assert abs(float(opt.grad_norm) - norm64.item()) <= 1e-5 * norm64.item()
    for k, p in model.named_parameters():
        if k not in g64:
            continue
        # Adam's first step is lr * g / (|g| + 1e-8): where |g| is not far above 1e-8 the step depends on
        # the gradient's last bits, so the values are compared where the gradient is clear of that
        clear = g64[k].abs() > 1e-5
        if not bool(clear.any()):
            continue
        want, got_p = p64[k].detach()[clear], p.detach().cpu().double()[clear]
        err = (got_p - want).abs().max().item()
        assert err <= 1e-5 * max(want.abs().max().item(), 1e-3), f'{k}: {err:.3e} vs values up to {want.abs().max().item():.3e}'


def test_infer_links_against_reference_golden(tokenizer):
    gold = load_npz('infer_links.npz')
    state = {k: torch.from_numpy(v) for k, v in sub(gold, 'state').items()}
    model, _ = hip_model(tokenizer, state=state, **SMALL)
    th = float(gold['add_edge_th'])
    for c in range(int(gold['num_cases'])):
        g = sub(gold, f'case{c}')
        groups, p = [], 0
        for ln in g['groups_len']:
            groups.append(g['groups_flat'][p:p + ln].tolist())
            p += ln
        pairs, conf, etype, order = model.decoder.infer_links(
            torch.from_numpy(g['z']), torch.from_numpy(g['x']), torch.from_numpy(g['x_pieces']),
            torch.from_numpy(g['x_pos']), torch.from_numpy(g['edge_index']), torch.from_numpy(g['edge_attr']),
            groups, th)
        assert np.array_equal(pairs.cpu().numpy(), g['pairs'])
        close(conf, torch.from_numpy(g['conf']), what='confidence')
        assert np.array_equal(etype.cpu().numpy(), g['etype'])
        # the kept set and its order agree wherever confidences are not within rounding of
        # each other or of the threshold
        want = g['order'].tolist()
        gc = g['conf']
        safe = all(abs(gc[i] - th) > 1e-5 for i in range(len(gc)))
        srt = np.sort(gc[want])[::-1] if len(want) else np.zeros(0)
        distinct = len(want) < 2 or float(np.min(-np.diff(srt))) > 1e-6
        if safe and distinct:
            assert order == want
        else:
            assert sorted(order) == sorted(want) or not safe


def test_sample_pieces_batch_against_oracle(tokenizer):
    """Batched piece sampling = the oracle's per-latent loop on the same uniforms (ids exact).
    No reference fixture covers sampling (torch.multinomial's own stream): parity unpinned."""
    model, cfg = hip_model(tokenizer, seed=77, **SMALL)
    ref = model_ref.PSVAEModel(cfg, tokenizer)
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=True)
    B, max_pos = 48, model.decoder.pos_embedding.num_embeddings
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, SMALL['latent_dim'], generator=g) * 2.0
    u = torch.rand(B, max_pos, generator=g)
    counts = (torch.arange(tokenizer.num_piece_type()) % 6 + 1).tolist()
    counts[tokenizer.end_idx()] = 0
    counts[tokenizer.pad_idx()] = 0
    for max_atoms, temp in ((20, 1.0), (60, 0.7), (10 ** 6, 1.3)):
        ids, lengths = model.decoder.sample_pieces_batch(z, max_atoms, temp, uniforms=u, piece_atom_counts=counts)
        ids, lengths = ids.cpu(), lengths.cpu()
        with torch.no_grad():
            for b in range(B):
                want = ref.decoder.sample_pieces(z[b], max_atoms, temp, u[b], counts)
                assert lengths[b].item() == len(want), (b, lengths[b].item(), len(want))
                assert ids[b, :len(want)].tolist() == want
                assert (ids[b, len(want):] == tokenizer.pad_idx()).all()


def test_infer_links_batch_equals_per_molecule(tokenizer):
    """The batched link prediction returns, molecule by molecule, what infer_links returns for
    that molecule alone (which the reference golden pins), on the golden cases run as ONE batch
    and on 64 synthetic molecules."""
    from ps_vae_amd import ops as hops
    from ps_vae_amd.data.synthetic import synthetic_step1
    gold = load_npz('infer_links.npz')
    state = {k: torch.from_numpy(v) for k, v in sub(gold, 'state').items()}
    model, _ = hip_model(tokenizer, state=state, **SMALL)
    dec = model.decoder
    th = float(gold['add_edge_th'])
    mols = []
    for c in range(int(gold['num_cases'])):
        g = sub(gold, f'case{c}')
        groups, p = [], 0
        for ln in g['groups_len']:
            groups.append(g['groups_flat'][p:p + ln].tolist())
            p += ln
        mols.append(dict(z=torch.from_numpy(g['z']).reshape(-1), x=torch.from_numpy(g['x']).reshape(-1),
                         xp=torch.from_numpy(g['x_pieces']).reshape(-1), xq=torch.from_numpy(g['x_pos']).reshape(-1),
                         ei=torch.from_numpy(g['edge_index']), ea=torch.from_numpy(g['edge_attr']), groups=groups))
    rng = np.random.default_rng(5)
    latent = mols[0]['z'].numel()
    for d in synthetic_step1(64, seed=11, shape='zinc'):
        n = len(d['x'])
        frag = np.zeros(n, dtype=np.int64)
        for gi, grp in enumerate(d['groups']):
            frag[grp] = gi
        bonds = [(a, b, c) for (a, b), c in zip(d['edge_index'], d['edge_attr']) if frag[a] == frag[b]]
        ei = torch.tensor([[a for a, b, c in bonds] + [b for a, b, c in bonds],
                           [b for a, b, c in bonds] + [a for a, b, c in bonds]], dtype=torch.int64).reshape(2, -1)
        cls = torch.tensor([c for a, b, c in bonds] * 2, dtype=torch.int64)
        ea = torch.nn.functional.one_hot(cls, 4).to(torch.int64) if len(bonds) else torch.zeros(0, 4, dtype=torch.int64)
        pieces = np.asarray(d['pieces'])
        mols.append(dict(z=torch.from_numpy(rng.standard_normal(latent).astype(np.float32)),
                         x=torch.tensor(d['x']), xp=torch.from_numpy(pieces[frag]), xq=torch.from_numpy(frag),
                         ei=ei, ea=ea, groups=[g for g in d['groups'] if len(g)]))
    # one flat batch
    node_off = np.concatenate([[0], np.cumsum([m['x'].numel() for m in mols])])
    cat = lambda key: torch.cat([m[key] for m in mols])
    frag_all = []
    for mol in mols:
        f = torch.full((mol['x'].numel(),), -1, dtype=torch.int64)
        for gi, grp in enumerate(mol['groups']):
            f[torch.as_tensor(grp, dtype=torch.int64)] = gi
        assert (f >= 0).all()
        frag_all.append(f)
    ei_all = torch.cat([m['ei'] + int(o) for m, o in zip(mols, node_off[:-1])], dim=1)
    ea_all = torch.cat([m['ea'] for m in mols])
    V = int(node_off[-1])
    csr = hops.Csr.from_edge_index(ei_all.to(dev()), ea_all.to(dev()), V)
    res = dec.infer_links_batch(torch.stack([m['z'] for m in mols]), cat('x'), cat('xp'), cat('xq'),
                                torch.from_numpy(node_off), csr, torch.cat(frag_all), th)
    ptr, optr = res['ptr'].cpu().numpy(), res['order_ptr'].cpu().numpy()
    for b, mol in enumerate(mols):
        pairs, conf, etype, order = dec.infer_links(mol['z'], mol['x'], mol['xp'], mol['xq'], mol['ei'], mol['ea'],
                                                    mol['groups'], th)
        lo, hi = ptr[b], ptr[b + 1]
        assert hi - lo == pairs.shape[0]
        assert np.array_equal(torch.stack([res['m'][lo:hi], res['n'][lo:hi]], 1).cpu().numpy(), pairs.cpu().numpy())
        close(res['conf'][lo:hi], conf, what=f'molecule {b} confidence')
        assert np.array_equal(res['etype'][lo:hi].cpu().numpy(), etype.cpu().numpy())
        got = (res['order'][optr[b]:optr[b + 1]] - int(lo)).cpu().tolist()
        c = conf.cpu().numpy()
        safe = all(abs(v - th) > 1e-5 for v in c)
        srt = np.sort(c[order])[::-1] if len(order) else np.zeros(0)
        distinct = len(order) < 2 or float(np.min(-np.diff(srt))) > 1e-6
        if safe and distinct:
            assert got == order
        elif safe:
            assert sorted(got) == sorted(order)


def test_assemble_links_batch_follows_the_greedy_loop(tokenizer):
    """infer_links_batch -> assemble_links_batch on 96 synthetic fragment sets: with the
    candidates, types and visiting order the GPU produced, the accepted bonds, final blocks and
    kept atoms equal the per-molecule Python loop of the oracle."""
    from oracle import assemble_ref
    from ps_vae_amd import ops as hops
    from ps_vae_amd.data.synthetic import synthetic_step1
    model, _ = hip_model(tokenizer, seed=3, **SMALL)
    dec = model.decoder
    rng = np.random.default_rng(8)
    latent = dec.latent_dim
    mols = []
    for d in synthetic_step1(96, seed=21, shape='zinc'):
        n = len(d['x'])
        groups = [g for g in d['groups'] if len(g)]
        frag = np.zeros(n, dtype=np.int64)
        for gi, grp in enumerate(groups):
            frag[grp] = gi
        bonds = [(a, b, c) for (a, b), c in zip(d['edge_index'], d['edge_attr']) if frag[a] == frag[b]]
        mols.append(dict(x=np.asarray(d['x']), frag=frag, groups=groups, bonds=bonds,
                         xp=rng.integers(0, 300, size=n)))
    node_off = np.concatenate([[0], np.cumsum([len(m['x']) for m in mols])])
    V = int(node_off[-1])
    src, dst, cls = [], [], []
    for m, o in zip(mols, node_off[:-1]):
        for a, b, c in m['bonds']:
            src += [o + a, o + b]
            dst += [o + b, o + a]
            cls += [c, c]
    ei = torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1)
    ea = torch.nn.functional.one_hot(torch.tensor(cls, dtype=torch.int64), 4).to(torch.int64)
    csr = hops.Csr.from_edge_index(ei.to(dev()), ea.to(dev()), V)
    x = torch.from_numpy(np.concatenate([m['x'] for m in mols]))
    frag = torch.from_numpy(np.concatenate([m['frag'] for m in mols]))
    xp = torch.from_numpy(np.concatenate([m['xp'] for m in mols]))
    z = torch.from_numpy(rng.standard_normal((len(mols), latent)).astype(np.float32))
    links = dec.infer_links_batch(z, x, xp, frag, torch.from_numpy(node_off), csr, frag, 0.3)
    n_blocks = torch.tensor([len(m['groups']) for m in mols])
    res = dec.assemble_links_batch(links, x, torch.from_numpy(node_off), csr, frag, n_blocks)
    assert res['status'].cpu().tolist() == [0] * len(mols)
    ptr, optr = links['ptr'].cpu().numpy(), links['order_ptr'].cpu().numpy()
    lm, ln, et = links['m'].cpu().numpy(), links['n'].cpu().numpy(), links['etype'].cpu().numpy()
    order = links['order'].cpu().numpy()
    acc, blk, keep = res['accepted'].cpu().numpy(), res['block'].cpu().numpy(), res['keep_atom'].cpu().numpy()
    n_acc = 0
    for b, m in enumerate(mols):
        lo, hi = int(ptr[b]), int(ptr[b + 1])
        cand = list(zip(lm[lo:hi].tolist(), ln[lo:hi].tolist()))
        srt = (order[optr[b]:optr[b + 1]] - lo).tolist()
        w_acc, w_blk, w_keep = assemble_ref.assemble(m['x'].tolist(), m['bonds'], m['groups'], cand,
                                                     et[lo:hi].tolist(), srt, tokenizer.chem_vocab)
        assert acc[optr[b]:optr[b + 1]].tolist() == w_acc, b
        a0, a1 = int(node_off[b]), int(node_off[b + 1])
        assert blk[a0:a1].tolist() == w_blk, b
        assert keep[a0:a1].tolist() == w_keep, b
        n_acc += sum(w_acc)
    assert n_acc > 0


def test_process_step3_same_seed_as_reference(tokenizer):
    """BPEMolDataset.process_step3 draws random.random() in the reference's order: under the
    seed the golden batch was made with it must return that batch."""
    import json
    import os
    import random
    from conftest import GOLDEN
    from ps_vae_amd.data.bpe_dataset import BPEMolDataset
    with open(os.path.join(GOLDEN, 'collate_cases.json')) as f:
        cases = json.load(f)
    gold_all = load_npz('collate_cases.npz')
    for i, (name, step1) in enumerate(cases.items()):
        gold = sub(gold_all, name)
        random.seed(100 + i)
        step2 = [BPEMolDataset.process_step2(d, tokenizer) for d in step1]
        got = BPEMolDataset.process_step3(step2, tokenizer, device=dev())
        for k in ('edge_select', 'golden_edge', 'edge_attr', 'in_piece_edge_idx', 'x_pieces', 'x_pos'):
            assert np.array_equal(got[k].cpu().numpy(), gold[k]), (name, k)
