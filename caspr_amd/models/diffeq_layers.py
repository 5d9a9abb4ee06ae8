"""ConcatSquashLinear parameter container (reference: caspr/models/diffeq_layers.py:76-90).

out = (W x + b) * sigmoid(W_g [t, c] + b_g) + W_b [t, c]; evaluated inside caspr_cnf_rk4_f32."""
import torch.nn as nn


class ConcatSquashLinear(nn.Module):
    def __init__(self, dim_in, dim_out, dim_c):
        super(ConcatSquashLinear, self).__init__()
        self._layer = nn.Linear(dim_in, dim_out)
        self._hyper_bias = nn.Linear(1 + dim_c, dim_out, bias=False)
        self._hyper_gate = nn.Linear(1 + dim_c, dim_out)
