"""Data-parallel gradient exchange: one process per GPU, one flat fp32 gradient buffer, one
all-reduce (RCCL over xGMI on the GPUs, gloo on CPU) per step.

The reference trains PS-VAE with single-process nn.DataParallel at most
(src/train.py:49-50); the only torch.distributed use in its tree is the GuacaMol predictor
trainer (src/guacamol_exps/goal_directed/train.py:159-161, 268-276).  Semantics here are
standard DDP: every rank computes the gradient of its own batch's loss (means over the
rank-local pair / token / molecule counts) and the ranks' gradients are averaged.

All parameter gradients are views into one contiguous buffer, so the exchange is a single
collective of 7.57 MB (1 891 727 fp32) with no bucketing, packing or unpacking kernels.
Parameters that never receive a gradient (decoder.graph_embedding.linear.*, 1 201 elements)
simply stay zero in the buffer and are never waited on."""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, parameters, process_group=None):
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dtype = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1

    def zero_grad(self):
        """Instead of optimizer.zero_grad(set_to_none=True), which would detach the views."""
        self.flat.zero_()

    def sync(self):
        """Average the gradients over the ranks (no-op for a single process)."""
        if self.world > 1:
            if self.flat.is_cuda and dist.get_backend(self.group) == 'gloo':
                # rehearsals of the multi-rank path without RCCL: stage through host memory
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world)

    def clip_grad_norm_(self, max_norm):
        """Global-norm clipping on the flat buffer (torch.nn.utils.clip_grad_norm_ semantics:
        scale by max_norm / (norm + 1e-6) when that is below 1).  Returns the norm tensor."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total
