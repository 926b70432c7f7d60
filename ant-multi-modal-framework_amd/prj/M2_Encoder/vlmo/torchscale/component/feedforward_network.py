"""FeedForwardNetwork parameter holder: fc1 -> gelu -> ffn_layernorm (sub-LN over 4d) -> fc2 (reference:
prj/M2_Encoder/vlmo/torchscale/component/feedforward_network.py:89-128)."""
from torch import nn


class FeedForwardNetwork(nn.Module):
    def __init__(self, embed_dim, ffn_dim, activation_fn, dropout, activation_dropout, layernorm_eps, subln=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.fc1 = nn.Linear(embed_dim, ffn_dim)
        self.fc2 = nn.Linear(ffn_dim, embed_dim)
        self.ffn_layernorm = nn.LayerNorm(ffn_dim, eps=layernorm_eps) if subln else None

    def reset_parameters(self):
        self.fc1.reset_parameters()
        self.fc2.reset_parameters()
        if self.ffn_layernorm is not None:
            self.ffn_layernorm.reset_parameters()
