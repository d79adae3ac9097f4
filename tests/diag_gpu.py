"""Diagnostic (not a test): per-parameter gradient error of the HIP model vs the float64
oracle next to the fp32 CPU oracle's own error, at default widths.  Run on the GPU box:
    python tests/diag_gpu.py [B]"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ps_vae_amd  # noqa: E402,F401
from conftest import Tokenizer  # noqa: E402
from oracle import collate_ref, model_ref  # noqa: E402
from ps_vae_amd.data.collate import collate_packed  # noqa: E402
from ps_vae_amd.data.packed import PackedMolecules  # noqa: E402
from ps_vae_amd.data.synthetic import synthetic_step1  # noqa: E402
from ps_vae_amd.pl_models import PSVAEModel  # noqa: E402
from ps_vae_amd.utils.nn_utils import build_config, default_args  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tok = Tokenizer()
dev = torch.device('cuda:0')
step1 = synthetic_step1(B, seed=123, shape='zinc')
uniforms = np.random.default_rng(1).random(700 * B)
want_batch, used = collate_ref.collate(step1, uniforms, piece_pad=tok.pad_idx())
torch.manual_seed(7)
ref = model_ref.PSVAEModel(model_ref.default_config(tok), tok)
ref64 = copy.deepcopy(ref).double()
model = PSVAEModel(build_config(default_args(), tok), tok)
model.load_state_dict(ref.state_dict())
model.to(dev)
eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(2))
tb = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in want_batch.items()}

acts = {}


def tap(store, name):
    def hook(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        store[name] = o.detach()
        if o.requires_grad:
            o.register_hook(lambda g: store.__setitem__(name + '.grad', g.detach()))
    return hook


def taps(m, store):
    m.decoder.graph_embedding.node_trans.register_forward_hook(tap(store, 'dec.node_trans'))
    m.encoder.node_trans.register_forward_hook(tap(store, 'enc.node_trans'))
    m.decoder.edge_predictor[0].register_forward_hook(tap(store, 'pair.mlp'))
    m.decoder.W_mean.register_forward_hook(tap(store, 'W_mean'))


a32, a64, agpu = {}, {}, {}
taps(ref, a32), taps(ref64, a64)
pl, (rec, piece, edge, kl) = ref(tb, eps)
ref.weighted_loss(pl, rec, kl).backward()
pl, (rec, piece, edge, kl) = ref64(tb, eps.double())
ref64.weighted_loss(pl, rec, kl).backward()

packed = PackedMolecules.from_step1(step1)
mode = os.environ.get('DIAG_PATH', 'compact')
if mode == 'compact':
    batch = collate_packed(packed.to_device(dev), np.arange(B), uniforms=torch.from_numpy(uniforms).to(dev),
                           atom_pad=tok.atom_pad_idx(), piece_pad=tok.pad_idx(), reference_layout=False)
else:
    batch = {k: v.to(dev) for k, v in tb.items()}
from ps_vae_amd import ops as _ops  # noqa: E402
import ps_vae_amd.modules.common_nn as _cn  # noqa: E402
_rec = []
_orig_linear = _ops.linear


def _rec_linear(x, w, b=None, relu=False):
    out = _orig_linear(x, w, b, relu=relu)
    _rec.append((tuple(w.shape), relu, out.detach()))
    return out


_ops.linear = _rec_linear
h1_64 = {}
ref64.decoder.edge_predictor[0].seq[1].register_forward_hook(lambda m, i, o: h1_64.__setitem__('h1', o.detach()))
ref64.decoder.edge_predictor[0].seq[3].register_forward_hook(lambda m, i, o: h1_64.__setitem__('h2', o.detach()))
ref.decoder.edge_predictor[0].seq[1].register_forward_hook(lambda m, i, o: h1_64.__setitem__('h1_32', o.detach()))
ref64(tb, eps.double())
ref(tb, eps)
gpl, (grec, gpiece, gedge, gkl) = model(batch, eps=eps.to(dev))
_ops.linear = _orig_linear
for shp, relu, out in _rec:
    if shp == (328, 656):
        a, b = out.cpu(), h1_64['h1']
        print('pair layer0: rel err', ((a.double() - b).abs().max() / b.abs().max()).item(),
              'sign flips GPU', ((a > 0) != (b > 0)).sum().item(),
              'sign flips CPU32', ((h1_64['h1_32'] > 0) != (b > 0)).sum().item(), 'of', b.numel(),
              'frac active', (b > 0).double().mean().item())
    if shp == (328, 328):
        a, b = out.cpu(), h1_64['h2']
        print('pair layer1: rel err', ((a.double() - b).abs().max() / b.abs().max()).item(),
              'sign flips GPU', ((a > 0) != (b > 0)).sum().item())
model.weighted_loss(gpl, grec, gkl).backward()

p32, p64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
print(f'{"parameter":52s} {"gpu_err":>10s} {"cpu32_err":>10s} {"scale":>10s}')
for k, p in model.named_parameters():
    if p64[k].grad is None:
        continue
    g64 = p64[k].grad
    s = g64.abs().max().item()
    eg = (p.grad.cpu().double() - g64).abs().max().item() / s
    ec = (p32[k].grad.double() - g64).abs().max().item() / s
    flag = ' <<<' if eg > max(1e-5, 2 * ec) else ''
    print(f'{k:52s} {eg:10.2e} {ec:10.2e} {s:10.2e}{flag}')

# ---- focused check of the pair MLP chain on identical inputs
from ps_vae_amd import ops  # noqa: E402
P = a64['pair.mlp'].shape[0]
g = torch.Generator().manual_seed(0)
feats = torch.randn(P, 656, generator=g)
gout = torch.randn(P, 656, generator=g) * 1e-3
mlp64 = ref64.decoder.edge_predictor[0]
f64 = feats.double().requires_grad_()
h1 = torch.relu(mlp64.seq[0](f64)); h1.retain_grad()
h2 = torch.relu(mlp64.seq[2](h1)); h2.retain_grad()
h3 = mlp64.seq[4](h2)
for p_ in mlp64.parameters():
    p_.grad = None
h3.backward(gout.double())
mg = model.decoder.edge_predictor[0]
for p_ in mg.parameters():
    p_.grad = None
fd = feats.to(dev).requires_grad_()
d1 = ops.linear(fd, mg.seq[0].weight, mg.seq[0].bias, relu=True); d1.retain_grad()
d2 = ops.linear(d1, mg.seq[2].weight, mg.seq[2].bias, relu=True); d2.retain_grad()
d3 = ops.linear(d2, mg.seq[4].weight, mg.seq[4].bias)
d3.backward(gout.to(dev))


def rel(a, b):
    return ((a.detach().cpu().double() - b.detach()).abs().max() / b.detach().abs().max()).item()


print('P', P)
for name, a, b in [('h1', d1, h1), ('h2', d2, h2), ('h3', d3, h3), ('g_h2', d2.grad, h2.grad), ('g_h1', d1.grad, h1.grad),
                   ('g_feats', fd.grad, f64.grad), ('gW0', mg.seq[0].weight.grad, mlp64.seq[0].weight.grad),
                   ('gb0', mg.seq[0].bias.grad, mlp64.seq[0].bias.grad),
                   ('gW2', mg.seq[2].weight.grad, mlp64.seq[2].weight.grad),
                   ('gW4', mg.seq[4].weight.grad, mlp64.seq[4].weight.grad)]:
    print(f'{name:8s} {rel(a, b):.2e}')
# mask flips between fp32 and fp64 forward
flip1 = ((d1.detach().cpu() > 0) != (h1.detach() > 0)).sum().item()
flip2 = ((d2.detach().cpu() > 0) != (h2.detach() > 0)).sum().item()
print('relu sign flips layer1/layer2:', flip1, flip2, 'of', h1.numel())
