"""MovingBatchNorm1d of the point CNF.  Behavioural contract = caspr/models/normalization.py:12-127:

    forward : y = (x - mu) / sqrt(var + eps) * exp(w) + b        logp -= sum_d (w_d - 0.5 log(var_d + eps))
    reverse : x = (y - b) * exp(-w) * sqrt(var + eps) + mu       logp += the same sum
    mu, var = running statistics taken BEFORE this call's update; in train mode the forward direction then
    moves them towards the batch statistics with decay 0.1 and increments `step`.  eps = 1e-4.

In the sampling / NLL path the transform runs inside caspr_cnf_rk4_f32's prologue and epilogue
(`kernel_params()` hands the 12 numbers over); `forward` below is the same map in torch for stand-alone use.
State-dict surface (Appendix C): weight, bias, step, running_mean, running_var."""
import torch
import torch.nn as nn

__all__ = ['MovingBatchNorm1d']


class MovingBatchNorm1d(nn.Module):
    def __init__(self, num_features, eps=1e-4, decay=0.1, affine=True):
        super().__init__()
        if not affine:
            raise ValueError("affine=False is not used by the reference flow (flow.py:68-72) and not supported")
        self.num_features, self.eps, self.decay, self.affine = num_features, eps, decay, affine
        self.weight = nn.Parameter(torch.zeros(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('step', torch.zeros(1))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))

    def reset_parameters(self):
        with torch.no_grad():
            self.weight.zero_()
            self.bias.zero_()
            self.running_mean.zero_()
            self.running_var.fill_(1)

    @torch.no_grad()
    def update_running_mean(self, x):
        """Exponential moving statistics, grouped exactly as the reference groups them (normalization.py:43-49):
        it swaps the first two axes and then RESHAPES to (num_features, -1), so for a (B, N, 3) cloud each
        "channel" statistic is taken over a contiguous third of the (N, B, 3) buffer, not over one coordinate.
        Reproduced on purpose: checkpoints were trained with these statistics."""
        chunks = x.transpose(0, 1).reshape(x.shape[-1], -1)
        self.running_mean.lerp_(chunks.mean(dim=1), self.decay)
        self.running_var.lerp_(chunks.var(dim=1), self.decay)    # unbiased, as torch.var in the reference
        self.step += 1

    def kernel_params(self):
        """[weight | bias | running_mean | running_var] as one 12-float device vector for the CNF kernel."""
        return torch.cat([self.weight.detach(), self.bias.detach(), self.running_mean, self.running_var]).float().contiguous()

    def log_scale(self):
        """Per-dimension log |dy/dx| of the forward map."""
        return self.weight - 0.5 * torch.log(self.running_var.detach().clone() + self.eps)

    def forward(self, x, context=None, logpx=None, integration_times=None, reverse=False):
        mu = self.running_mean.detach().clone()
        log_s = self.log_scale().detach() if not self.weight.requires_grad else self.weight - 0.5 * torch.log(self.running_var.detach().clone() + self.eps)
        if reverse:
            out = (x - self.bias) * torch.exp(-log_s) + mu
            new_logp = None if logpx is None else logpx + log_s.sum().expand_as(logpx)
        else:
            if self.training:
                self.update_running_mean(x)      # after mu / log_s were taken: this call still uses the old statistics
            out = (x - mu) * torch.exp(log_s) + self.bias
            new_logp = None if logpx is None else logpx - log_s.sum().expand_as(logpx)
        return out if logpx is None else (out, new_logp)

    def extra_repr(self):
        return '{num_features}, eps={eps}, decay={decay}, affine={affine}'.format(**self.__dict__)
