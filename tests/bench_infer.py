"""Micro-benchmark (not a test): batched link prediction over all candidate atom pairs of N
generated molecules (VAEPieceDecoder.infer_links_batch), synthetic ZINC-shaped fragments.
    python tests/bench_infer.py [N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd.data.bpe_dataset import BPEMolDataset  # noqa: E402
from ps_vae_amd.data.packed import PackedMolecules  # noqa: E402
from ps_vae_amd.data.synthetic import synthetic_step1  # noqa: E402
from ps_vae_amd.pl_models import PSVAEModel  # noqa: E402
from ps_vae_amd.utils.nn_utils import build_config, default_args  # noqa: E402
from ps_vae_amd.utils.tokenizer import Tokenizer  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device('cuda:0')
tok = Tokenizer()
torch.manual_seed(2021)
model = PSVAEModel(build_config(default_args(), tok), tok).to(dev)
data = BPEMolDataset(None, tok, device=dev, data=PackedMolecules.from_step1(synthetic_step1(N, seed=7, shape='zinc')))
batch = data.collate_ids(np.arange(N))
c = batch.compact
z = torch.randn(N, model.decoder.latent_dim, device=dev)


def run():
    return model.decoder.infer_links_batch(z, c.cx, c.cxp, c.cxpos, c.node_off, c.csr_in, c.cxpos, 0.5)


res = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    res = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
P = res['conf'].numel()
print(f'{N} molecules, {c.num_nodes} atoms, {P} candidate pairs, {res["order"].numel()} kept: '
      f'{dt * 1e3:.1f} ms per batch = {N / dt:.0f} molecules/s, {P / dt / 1e6:.1f} M pairs/s')
