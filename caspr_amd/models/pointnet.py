"""Global spatio-temporal PointNet (reference: caspr/models/pointnet.py:18-46), on the HIP kernels.

Same parameter tree as the reference's `PointNetfeat` (conv1..3, bn1..3).  The reference tiles the
1024-channel max feature over all T*N points and concatenates it with the 64-channel point feature
(pointnet.py:44-46, 1.43 GB at B=16,T=10,N=2048); here `features()` returns the two pieces and the
consumer (TPointNet2 head) folds the max feature into a per-sequence bias instead.
"""
import torch
import torch.nn as nn

from .. import ops
from ..utils.weight_cache import WeightCache
from .lazy import Lazy


class PointNetfeat(nn.Module):
    def __init__(self, input_dim=3, out_size=1024, layer_sizes=[64, 128]):
        super(PointNetfeat, self).__init__()
        self.output_size = out_size
        self.input_dim = input_dim
        self.conv1 = torch.nn.Conv1d(self.input_dim, layer_sizes[0], 1)
        self.conv2 = torch.nn.Conv1d(layer_sizes[0], layer_sizes[1], 1)
        self.conv3 = torch.nn.Conv1d(layer_sizes[1], self.output_size, 1)
        self.bn1 = nn.GroupNorm(16, layer_sizes[0])
        self.bn2 = nn.GroupNorm(16, layer_sizes[1])
        self.bn3 = nn.GroupNorm(16, self.output_size)
        self._cache = WeightCache()

    def _packed(self, name):
        conv = getattr(self, name)
        return self._cache.get(name, [conv.weight], lambda: ops.PackedWeight(conv.weight.detach()[:, :, 0].contiguous()))

    def features(self, x_pm, y1_out=None):
        """x_pm (B,P,4) point-major.  Returns (pointfeat Lazy (B,P,64) = relu(bn1(conv1)), gmax (B,1024)).
        y1_out: optional (B,P,64) column slice to hold conv1's raw output."""
        c1, c2, c3 = self.conv1, self.conv2, self.conv3
        y1 = ops.conv1x1(self._packed("conv1"), c1.bias, x_pm, out=y1_out)                     # pointnet.py:37
        s1, t1 = ops.gn_stats(y1, c1.out_channels, self.bn1.weight, self.bn1.bias)
        y2, s2, t2 = ops.conv1x1_gn(self._packed("conv2"), c2.bias, y1, self.bn2.weight, self.bn2.bias,
                                    in_scale=s1, in_shift=t1, in_relu=True)                     # :39, statistics in the conv's epilogue
        # conv3 -> bn3 -> max over points (:40-42): only the pooled maximum is used, so the 1024-channel output is not stored
        _, _, _, gmax = ops.conv1x1_gn(self._packed("conv3"), c3.bias, y2, self.bn3.weight, self.bn3.bias, want_max=True, write=False,
                                       in_scale=s2, in_shift=t2, in_relu=True)
        return Lazy(y1, c1.out_channels, s1, t1, True), gmax

    def forward(self, x):
        """Reference signature: x (B,input_dim,P) channels-first -> (B, out_size + 64, P)."""
        n_pts = x.size()[2]
        pf, gmax = self.features(x.transpose(1, 2).contiguous())
        return torch.cat([gmax.unsqueeze(2).repeat(1, 1, n_pts), pf.materialize().transpose(1, 2)], 1)
