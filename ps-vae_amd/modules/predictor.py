"""Property predictor on z with the reference's layout (src/modules/predictor.py:6-21)."""
import torch.nn as nn

from .common_nn import run_linear, run_sequential


class Predictor(nn.Module):
    def __init__(self, dim_feature, dim_hidden, num_property):
        super(Predictor, self).__init__()
        self.mlp = nn.Sequential(
            nn.Linear(dim_feature, dim_hidden), nn.ReLU(),
            nn.Linear(dim_hidden, dim_hidden), nn.ReLU())
        self.output = nn.Linear(dim_hidden, num_property)

    def forward(self, x):
        return run_linear(self.output, run_sequential(self.mlp, x))
