"""The training step of small batches as ONE HIP-graph launch.

At batch 32 (the reference's own recipe, scripts/train.sh:14-36) a step is ~230 kernel launches of a
few microseconds each and the host cannot issue them as fast as the GPU finishes them.  Here the
step - forward, backward, gradient packing, global-norm clip and Adam
(src/pl_models/ps_vae_model.py:33-88, 123-127, src/train.py:47) - is recorded once into a HIP graph
over fixed-size buffers and replayed with one launch per step.  What changes from step to step is
data in device memory, never a kernel argument:

  * the batch: collation (data.collate.collate_static) refills the same padded buffers; rows beyond
    a batch's own atom / bond / pair counts are padding that takes no part in any result, the live
    pair count is read on the device by the link loss;
  * the KL weight (PSVAEModel.cal_beta) and Adam's two step-dependent scalars: written by the host
    into small device tensors before every replay.

A batch that does not fit the fixed sizes (a tail event, see StaticShapes.for_dataset) is stepped the
ordinary way.  In a multi-GPU run (one process per GPU) the recording ends where the gradients are
packed into the flat buffer; the all-reduce and the two optimiser kernels follow it as ordinary calls,
so every step has exactly one collective on every rank whichever way the rank stepped.
The recording keeps the model's two-stream fork (encoder beside decoder GNN + piece LM) as parallel
branches of the graph."""
import torch

from . import hip
from .data.collate import PinnedRing, StaticBatch, StaticShapes, collate_static


class GraphedTrainStep:
    def __init__(self, model, optimizer, dataset, batch_size, sigmas=4.0, shapes=None, warmup=3, generator=None):
        """model: PSVAEModel on a GPU; optimizer: optim.FlatAdam over its parameters (its max_norm
        is the clipping threshold); dataset: BPEMolDataset (packed molecules on the device)."""
        if not hasattr(optimizer, 'sync'):
            raise ValueError('GraphedTrainStep needs optim.FlatAdam')
        self.exchange_outside = optimizer.sync.world > 1
        self.model, self.opt, self.dataset = model, optimizer, dataset
        # The decoder's message passing and the piece LM go to the model's second stream whatever the batch
        # size: in a recording the fork costs the host nothing and the two branches run side by side
        # (measured at batch 32: 2.17 -> 1.67 ms per step ZINC-shaped, 1.72 -> 1.21 ms QM9-shaped).  The
        # warm-up steps fork as well, so that the second stream's workspaces exist before the recording.
        if model.two_streams:
            model.two_streams_min_nodes = 0
        self.device = model.device
        self.shapes = shapes or StaticShapes.for_dataset(dataset.packed, batch_size, sigmas)
        tok = dataset.tokenizer
        self.static = StaticBatch(self.shapes, self.device, tok.chem_vocab.num_bond_type())
        self.pads = dict(atom_pad=tok.atom_pad_idx(), piece_pad=tok.pad_idx())
        self.generator = generator
        self._host = PinnedRing((3,), torch.float32)                       # beta, step size, sqrt(1 - beta2^t)
        self._step_values = torch.zeros(3, dtype=torch.float32, device=self.device)   # one copy per step fills all three
        self.beta, self.scalars = self._step_values[0], self._step_values[1:]
        self.graph, self.loss = None, None
        self.warmup_left = max(1, warmup)   # at least one ordinary step: lazy set-up (cached index tensors, workspaces) cannot be recorded
        self.replays = self.fallbacks = 0

    # ---- pieces of one step
    def _scalars_to_device(self):
        group = self.opt.param_groups[0]
        step_size, bc2_sqrt = hip.adam_step_scalars(self.opt.steps + 1, group['lr'], group['betas'])
        host = self._host.next()
        host[0], host[1], host[2] = float(self.model.cal_beta()), step_size, bc2_sqrt
        self._step_values.copy_(host, non_blocking=True)
        self._host.sent()

    def _body(self, batch):
        """Forward, backward and the optimiser's kernels on `batch`, all step-dependent numbers taken
        from device memory."""
        self.opt.zero_grad()
        loss = self.model.training_step(batch, 0)
        loss.backward()
        if self.exchange_outside:
            self.opt.sync.pack()
        else:
            self.opt.step(step_scalars=self.scalars)
        return loss.detach()

    def _tail(self):
        """What follows the recording in a multi-GPU run: all-reduce, clip, Adam."""
        if self.exchange_outside:
            self.opt.step(step_scalars=self.scalars, packed=True)

    def _eager(self, ids):
        """The ordinary step for a batch beyond the fixed sizes."""
        self.fallbacks += 1
        batch = self.dataset.collate_ids(ids)
        self._scalars_to_device()
        loss = self._body(batch)
        self._tail()
        self.opt.steps += 1
        self.model.global_step += 1
        return loss

    def step(self, ids, uniforms=None):
        """One optimisation step on molecules `ids` of the dataset; returns the loss (a device scalar;
        in graph mode always the same tensor, rewritten by every replay)."""
        model = self.model
        if model.beta_tensor is None:
            model.beta_tensor = self.beta
        if len(ids) != self.shapes.B:          # the short last batch of an epoch
            return self._eager(ids)
        batch = collate_static(self.dataset.dpacked, ids, self.static, uniforms=uniforms, generator=self.generator,
                               **self.pads)
        if batch is None:
            return self._eager(ids)
        self._scalars_to_device()
        if self.graph is None and self.warmup_left > 0:
            # a few ordinary steps first: lazy set-up inside the libraries must not land in the recording
            self.warmup_left -= 1
            loss = self._body(batch)
        else:
            if self.graph is None:
                torch.cuda.synchronize(self.device)
                self.graph = torch.cuda.CUDAGraph()
                # (several ranks: the process group's watchdog thread may touch the GPU while this thread records)
                with torch.cuda.graph(self.graph, capture_error_mode='thread_local' if self.exchange_outside else 'global'):
                    self.loss = self._body(batch)
                # (recording runs nothing: the replay below is this step)
            self.graph.replay()
            self.replays += 1
            loss = self.loss
        self._tail()
        self.opt.steps += 1
        model.global_step += 1
        return loss
