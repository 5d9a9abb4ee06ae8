"""ODE function of the point CNF (reference: caspr/models/odefunc.py:62-142): parameter containers.
The arithmetic (4 gated layers, softplus, Hutchinson divergence) runs inside caspr_cnf_rk4_f32."""
import copy

import torch
import torch.nn as nn

from . import diffeq_layers

__all__ = ["ODEnet", "ODEfunc"]


class ODEnet(nn.Module):
    def __init__(self, hidden_dims, input_shape, context_dim, layer_type="concat", nonlinearity="softplus"):
        super(ODEnet, self).__init__()
        if layer_type != "concatsquash" or nonlinearity != "softplus":
            raise ValueError("the CNF kernel implements layer_type='concatsquash' with softplus (flow.py:91-92); got %s/%s" % (layer_type, nonlinearity))
        layers = []
        activation_fns = []
        hidden_shape = input_shape
        for dim_out in (hidden_dims + (input_shape[0],)):
            layers.append(diffeq_layers.ConcatSquashLinear(hidden_shape[0], dim_out, context_dim))
            activation_fns.append(nn.Softplus())
            hidden_shape = list(copy.copy(hidden_shape))
            hidden_shape[0] = dim_out
        self.layers = nn.ModuleList(layers)
        self.activation_fns = nn.ModuleList(activation_fns[:-1])


class ODEfunc(nn.Module):
    def __init__(self, diffeq):
        super(ODEfunc, self).__init__()
        self.diffeq = diffeq
        self.register_buffer("_num_evals", torch.tensor(0.))
        self._e = None
        self._count_evals = True      # False while the accuracy guard repeats a solve: the counter then keeps the real solve's value

    def before_odeint(self, e=None):
        """odefunc.py:115-117: fix (or clear) the Hutchinson noise for the next solve."""
        self._e = e
        if self._count_evals:
            self._num_evals.fill_(0)
