"""Data-parallel gradient exchange: one process per GPU, one flat fp32 gradient buffer, one
all-reduce (RCCL over xGMI on the GPUs, gloo on CPU) per step.

The reference trains PS-VAE with single-process nn.DataParallel at most
(src/train.py:49-50); the only torch.distributed use in its tree is the GuacaMol predictor
trainer (src/guacamol_exps/goal_directed/train.py:159-161, 268-276).  Semantics here are
standard DDP: every rank computes the gradient of its own batch's loss (means over the
rank-local pair / token / molecule counts) and the ranks' gradients are averaged.

After backward the gradients are packed into one contiguous buffer by a single multi-tensor
copy and every .grad becomes a view of it, so the exchange is a single collective of 7.57 MB
(1 891 727 fp32) and clipping / the optimiser work on the same memory.
Parameters that never receive a gradient (decoder.graph_embedding.linear.*, 1 201 elements)
simply stay zero in the buffer and are never waited on."""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, parameters, process_group=None, flatten_params=False):
        """flatten_params: also move the parameters themselves into one flat buffer
        (`flat_params`, every parameter becoming a view of it, values kept), so that an
        optimiser can update the whole model with one kernel (optim.FlatAdam).  Call it after
        the model sits on its device: `module.to()` would re-allocate the parameters."""
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dtype = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dtype for p in self.params):
            raise ValueError('parameters must share one device and dtype')
        # every parameter starts on a multiple of 4 elements: its rows stay 16-byte aligned
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, self.params)]
        with torch.no_grad():
            for p, v in zip(self.params, self.views):
                if p.grad is not None:        # gradients that already exist move into the buffer
                    v.copy_(p.grad)
                p.grad = v
        self.flat_params = None
        if flatten_params:
            self.flat_params = torch.zeros(total, dtype=dtype, device=dev)
            with torch.no_grad():
                for o, p in zip(offs, self.params):
                    slot = self.flat_params[o:o + p.numel()].view_as(p)
                    slot.copy_(p)
                    p.data = slot
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1

    def zero_grad(self):
        """Drop the gradients: backward then hands every parameter its gradient tensor as
        produced (no per-parameter accumulation kernels) and sync() packs them."""
        for p in self.params:
            p.grad = None

    def pack(self):
        """Copy the gradients backward produced into the flat buffer (one multi-tensor copy) and
        point every .grad at its slice; parameters without a gradient get zeros."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                src.append(g)
                dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)

    def sync(self, average=True):
        """Pack, then sum the gradients over the ranks and (average=True) divide by the world
        size; optim.FlatAdam passes False and folds the division into its update kernel.
        One process: pack only."""
        self.pack()
        self.exchange(average)

    def exchange(self, average=True):
        """The collective alone, on gradients that are already packed."""
        if self.world > 1:
            if self.flat.is_cuda and dist.get_backend(self.group) == 'gloo':
                # rehearsals of the multi-rank path without RCCL: stage through host memory
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if average:
                self.flat.div_(self.world)

    def clip_grad_norm_(self, max_norm):
        """Global-norm clipping on the flat buffer (torch.nn.utils.clip_grad_norm_ semantics:
        scale by max_norm / (norm + 1e-6) when that is below 1).  Returns the norm tensor."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total
