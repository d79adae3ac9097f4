import cProfile, pstats, sys, os, io
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import ps_vae_amd
from ps_vae_amd.pl_models import PSVAEModel
from ps_vae_amd.utils.nn_utils import build_config, default_args
from ps_vae_amd.utils.tokenizer import Tokenizer
from ps_vae_amd.data.bpe_dataset import BPEMolDataset
from ps_vae_amd.data.packed import PackedMolecules
from ps_vae_amd.data.synthetic import synthetic_step1
dev = torch.device('cuda:0')
tok = Tokenizer(); args = default_args()
torch.manual_seed(0)
model = PSVAEModel(build_config(args, tok), tok).to(dev)
(opt,), _ = model.configure_optimizers(); opt.max_norm = 10.0
B = 32
data = BPEMolDataset(None, tok, device=dev, data=PackedMolecules.from_step1(synthetic_step1(2048, seed=1, shape='zinc')))
rng = np.random.default_rng(0)
def step():
    batch = data.collate_ids(rng.choice(len(data), size=B, replace=False))
    opt.zero_grad()
    model.training_step(batch, 0).backward()
    opt.step(); model.global_step += 1
for _ in range(30): step()
torch.cuda.synchronize()
import time
t0=time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize()
print('ms/step', (time.perf_counter()-t0)*10)
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(35); print(s.getvalue()[:6000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40); print(s.getvalue()[:7000])
