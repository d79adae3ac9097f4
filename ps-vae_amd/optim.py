"""The optimiser tail on flat buffers: gradient exchange, global-norm clipping and Adam as two
kernel launches over the whole model (psvae_clip_adam_step), in place of the per-step sequence
of the reference's Trainer: clip_grad_norm_ (gradient_clip_val, src/train.py:47) and
torch.optim.Adam (src/pl_models/ps_vae_model.py:123-127)."""
import torch

from . import hip
from .dist import FlatGradSync


class FlatAdam(torch.optim.Optimizer):
    """Adam over a FlatGradSync whose parameters were flattened.  `step()` = pack the gradients
    backward produced, all-reduce them over the ranks, divide by the world size, clip to
    `max_norm` (None: no clipping) and update — the division, the clipping and the update being
    one kernel pair.  A torch.optim.Optimizer, so lr schedulers (LambdaLR of
    configure_optimizers) drive param_groups[0]['lr'] as usual."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=None,
                 process_group=None):
        sync = params if isinstance(params, FlatGradSync) else FlatGradSync(params, process_group,
                                                                             flatten_params=True)
        if sync.flat_params is None:
            raise ValueError('FlatAdam needs FlatGradSync(..., flatten_params=True)')
        if not sync.flat.is_cuda:
            raise RuntimeError('FlatAdam runs on the GPU (there is no CPU fallback)')
        super().__init__(sync.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.sync = sync
        self.max_norm = max_norm
        self.exp_avg = torch.zeros_like(sync.flat_params)
        self.exp_avg_sq = torch.zeros_like(sync.flat_params)
        self.steps = 0
        self.grad_norm = torch.zeros((), dtype=torch.float32, device=sync.flat.device)

    def zero_grad(self, set_to_none=True):
        self.sync.zero_grad()

    @torch.no_grad()
    def step(self, closure=None, step_scalars=None, packed=False):
        """step_scalars (device float32 [2] = hip.adam_step_scalars(step number, lr)): the update takes
        its step-dependent numbers from device memory and `self.steps` is left to the caller - the form
        a step recorded in a HIP graph needs (graph_step.GraphedTrainStep).  packed: the gradients
        already sit in the flat buffer (sync.pack() was part of a recording); only the collective and
        the update remain."""
        loss = closure() if closure is not None else None
        sync, group = self.sync, self.param_groups[0]
        if packed:
            sync.exchange(average=False)
        else:
            sync.sync(average=False)
        if step_scalars is None:
            self.steps += 1
        hip.clip_adam_step(sync.flat_params, sync.flat, self.exp_avg, self.exp_avg_sq, self.steps, group['lr'],
                           group['betas'], group['eps'], group['weight_decay'], grad_scale=1.0 / sync.world,
                           max_norm=self.max_norm, write_grad=True, norm_out=self.grad_norm, step_scalars=step_scalars)
        return loss

    def state_dict(self):
        """The layout of torch.optim.Adam's state_dict (what a checkpoint of the reference holds and
        torch.optim.Adam.load_state_dict reads): per parameter, in parameter order, {'step', 'exp_avg',
        'exp_avg_sq'} with the moments shaped like the parameter, and the param_groups with their
        parameter indices."""
        sync, state = self.sync, {}
        offset = 0
        for i, p in enumerate(sync.params):
            n = p.numel()
            state[i] = {'step': torch.tensor(float(self.steps)),
                        'exp_avg': self.exp_avg[offset:offset + n].view_as(p).clone(),
                        'exp_avg_sq': self.exp_avg_sq[offset:offset + n].view_as(p).clone()}
            offset += (n + 3) // 4 * 4
        groups = [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]
        groups[0]['params'] = list(range(len(sync.params)))
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, state):
        """Reads the layout above (also a state_dict written by torch.optim.Adam over the same parameters)
        and the flat layout earlier checkpoints of this package hold ({'steps', 'exp_avg', 'exp_avg_sq'})."""
        if 'state' in state:
            offset = 0
            steps = 0
            for i, p in enumerate(self.sync.params):
                n = p.numel()
                entry = state['state'].get(i)
                if entry is not None:
                    self.exp_avg[offset:offset + n].copy_(entry['exp_avg'].reshape(-1))
                    self.exp_avg_sq[offset:offset + n].copy_(entry['exp_avg_sq'].reshape(-1))
                    steps = max(steps, int(float(entry['step'])))
                else:                      # a parameter that never had a gradient has no entry in torch's dict
                    self.exp_avg[offset:offset + n].zero_()
                    self.exp_avg_sq[offset:offset + n].zero_()
                offset += (n + 3) // 4 * 4
            self.steps = steps
        else:
            self.steps = int(state['steps'])
            self.exp_avg.copy_(state['exp_avg'])
            self.exp_avg_sq.copy_(state['exp_avg_sq'])
        for g, saved in zip(self.param_groups, state['param_groups']):
            g.update({k: v for k, v in saved.items() if k != 'params'})
