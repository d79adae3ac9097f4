"""MLP with the reference's constructor, attribute (`seq`) and state_dict keys
(src/modules/common_nn.py:8-23); every Linear runs on the MFMA GEMM with bias and ReLU fused
into its epilogue."""
import torch.nn as nn

from .. import ops


def run_linear(layer, x, relu=False):
    return ops.linear(x, layer.weight, layer.bias, relu=relu)


def run_sequential(seq, x):
    """Apply an nn.Sequential of Linear / activation modules, fusing Linear + ReLU pairs."""
    layers = list(seq)
    i = 0
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, nn.Linear):
            fuse = i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU)
            x = run_linear(layer, x, relu=fuse)
            i += 2 if fuse else 1
        else:
            x = layer(x)
            i += 1
    return x


class MLP(nn.Module):
    def __init__(self, dim_in, dim_hidden, dim_out, act_func, num_layers):
        super(MLP, self).__init__()
        assert num_layers > 0
        if num_layers == 1:
            self.seq = nn.Linear(dim_in, dim_out)
        else:
            seq = [nn.Linear(dim_in, dim_hidden), act_func()]
            for _ in range(num_layers - 2):
                seq += [nn.Linear(dim_hidden, dim_hidden), act_func()]
            seq.append(nn.Linear(dim_hidden, dim_out))
            self.seq = nn.Sequential(*seq)

    def forward(self, x):
        if isinstance(self.seq, nn.Linear):
            return run_linear(self.seq, x)
        return run_sequential(self.seq, x)
