"""Not a test: a soak of the recorded batch-32 step on the real-molecule fixtures (tests/golden/real/): thousands of
steps, loss at the start and at the end, steps that fell back to the ordinary path, allocated device memory.
    python tests/bench_soak_b32.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd.graph_step import GraphedTrainStep  # noqa: E402
from ps_vae_amd.pl_models import PSVAEModel  # noqa: E402
from ps_vae_amd.utils.nn_utils import build_config, default_args  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dev = torch.device('cuda:0')
for shape in ('zinc_real', 'qm9_real'):
    tok = bench.real_tokenizer(shape[:-5])
    ds = bench.make_dataset(2048, 7, tok, dev, shape=shape)
    torch.manual_seed(2021)
    model = PSVAEModel(build_config(default_args(), tok), tok).to(dev)
    (opt,), _ = model.configure_optimizers()
    opt.max_norm = 10.0
    g = GraphedTrainStep(model, opt, ds, 32)
    rng = np.random.default_rng(3)
    losses, mem = [], []
    t0 = time.perf_counter()
    for i in range(steps):
        loss = g.step(rng.choice(len(ds), size=32, replace=False))
        if i % (steps // 8) == 0 or i == steps - 1:
            losses.append(round(float(loss), 4))
            mem.append(torch.cuda.memory_allocated(dev) >> 20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    print(f'{shape}: {steps} steps, {dt / steps * 1e3:.3f} ms per step wall (loss read 9 times), {g.replays} replays, '
          f'{g.fallbacks} fallbacks, loss {losses}, allocated MiB {mem}, parameters finite: {finite}', flush=True)
