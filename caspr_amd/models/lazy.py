"""`Lazy` activation: a raw conv output plus the per-(batch, channel) GroupNorm scale/shift (+ReLU)
that its consumers fold into their operand loads (conv1x1 / three_interpolate kernels), so
normalised activations are never written back to HBM."""
import torch


class Lazy:
    def __init__(self, raw, channels, scale=None, shift=None, relu=False):
        self.raw = raw            # (B, P, ld) point-major
        self.channels = channels
        self.scale = scale        # (B, C) or None
        self.shift = shift
        self.relu = relu

    def materialize(self):
        """Dense (B, P, C) tensor (used only at module boundaries that must return reference-shaped data)."""
        x = self.raw[:, :, :self.channels]
        if self.scale is not None:
            x = x * self.scale.unsqueeze(1) + self.shift.unsqueeze(1)
            if self.relu:
                x = torch.relu(x)
        return x.contiguous()
