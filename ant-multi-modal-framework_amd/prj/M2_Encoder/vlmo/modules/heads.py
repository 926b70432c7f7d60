"""Pooler / ITCHead (reference: prj/M2_Encoder/vlmo/modules/heads.py:4-24)."""
import torch
from torch import nn

from antmmf.hip import functional as HF


class Pooler(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        return torch.tanh(HF.linear(hidden_states[:, 0].contiguous(), self.dense.weight, self.dense.bias).float())


class ITCHead(nn.Module):
    def __init__(self, hidden_size, out_size):
        super().__init__()
        self.fc = nn.Linear(hidden_size, out_size, bias=False)

    def forward(self, x):
        return HF.linear(x.contiguous(), self.fc.weight)
