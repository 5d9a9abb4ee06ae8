"""Training tier: taped forward + explicit backward on the HIP gradient kernels (include/caspr_hip_train.h)."""
from .encoder_grad import EncoderFunction, encode_with_grad, encoder_forward, encoder_backward  # noqa: F401
