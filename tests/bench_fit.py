"""Not a test: throughput of train.fit - the loop a user of the reference's train.py runs - beside the bench
line's step.  Two configurations on synthetic ZINC-shaped molecules: batch 4096 (ordinary steps) and batch 32
(recorded steps); the first epoch warms up, the following ones are timed between device synchronisations.
    python tests/bench_fit.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ps_vae_amd  # noqa: E402,F401
from ps_vae_amd import train  # noqa: E402
from ps_vae_amd.data.bpe_dataset import _Loader  # noqa: E402
from ps_vae_amd.pl_models import PSVAEModel  # noqa: E402
from ps_vae_amd.utils.nn_utils import build_config, default_args  # noqa: E402
from ps_vae_amd.utils.tokenizer import Tokenizer  # noqa: E402

dev = torch.device('cuda:0')
tok = Tokenizer()
for batch, molecules, epochs, graph_steps in ((4096, 65536, 4, False), (32, 4096, 4, True)):
    ds = bench.make_dataset(molecules, 7, tok, dev, shape='zinc')
    torch.manual_seed(2021)
    model = PSVAEModel(build_config(default_args(), tok), tok).to(dev)
    loader = _Loader(ds, batch, True, reference_layout=False, seed=3)
    marks = []

    def on_step(m, loss, marks=marks, per_epoch=len(loader)):
        if m.global_step % per_epoch == 0:          # the end of an epoch
            torch.cuda.synchronize()
            marks.append(time.perf_counter())

    hist = train.fit(model, loader, max_epochs=epochs, on_step=on_step, graph_steps=graph_steps)
    timed = (marks[-1] - marks[0]) / (len(marks) - 1)
    print(f'fit batch {batch}: {len(loader)} steps per epoch, {timed / len(loader) * 1e3:.3f} ms per step, '
          f'{molecules / timed:,.0f} molecules/s over {len(marks) - 1} timed epochs '
          f'({hist.get("graph_replays", 0)} recorded steps replayed)', flush=True)
