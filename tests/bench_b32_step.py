"""Time of the recorded batch-32 training step (400 replays after 30 warm-up steps) on ZINC- and QM9-shaped
synthetic molecules.  Not a test: run as `python tests/bench_b32_step.py` on a GPU box."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
import ps_vae_amd, bench
from ps_vae_amd.pl_models import PSVAEModel
from ps_vae_amd.utils.nn_utils import build_config, default_args
from ps_vae_amd.utils.tokenizer import Tokenizer
from ps_vae_amd.graph_step import GraphedTrainStep
dev = torch.device('cuda:0'); tok = Tokenizer()
for shape in ('zinc', 'qm9'):
    ds = bench.make_dataset(2048, 7, tok, dev, shape=shape)
    torch.manual_seed(2021)
    m = PSVAEModel(build_config(default_args(), tok), tok).to(dev)
    (o,), _ = m.configure_optimizers(); o.max_norm = 10.0
    g = GraphedTrainStep(m, o, ds, 32)
    rng = np.random.default_rng(3)
    for _ in range(30): g.step(rng.choice(len(ds), size=32, replace=False))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(400): g.step(rng.choice(len(ds), size=32, replace=False))
    torch.cuda.synchronize(); print(shape, round((time.perf_counter() - t0) / 400 * 1e3, 3), 'ms/step', 'loss', float(g.loss))
