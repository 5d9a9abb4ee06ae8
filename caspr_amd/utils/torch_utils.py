"""Checkpoint helpers (reference: caspr/utils/torch_utils.py:27-65)."""
import numpy as np
import torch


def get_device():
    if not torch.cuda.is_available():
        raise RuntimeError("caspr_amd needs a ROCm GPU (MI355X); there is no CPU execution path")
    return torch.device('cuda:0')


def _strip_module_prefix(state_dict):
    for k in state_dict:
        if k.split('.')[0] == 'module':   # trained with DataParallel (torch_utils.py:32-35)
            return {'.'.join(k.split('.')[1:]): v for k, v in state_dict.items() if k.split('.')[0] == 'module'}
        break
    return state_dict


def load_weights(model, state_dict):
    """torch_utils.py:27-44: strict=False load, warns about missing / unexpected keys."""
    state_dict = _strip_module_prefix(state_dict)
    missing_keys, unexpected_keys = model.load_state_dict(state_dict, strict=False)
    if len(missing_keys) > 0:
        print('WARNING: The following keys could not be found in the given state dict - ignoring...')
        print(missing_keys)
    if len(unexpected_keys) > 0:
        print('WARNING: The following keys were found in the given state dict but not in the current model - ignoring...')
        print(unexpected_keys)


def load_encoder_weights_from_full(model, state_dict):
    """torch_utils.py:46-60."""
    state_dict = _strip_module_prefix(state_dict)
    state_dict = {'.'.join(k.split('.')[1:]): v for k, v in state_dict.items() if k.split('.')[0] == 'encoder'}
    model.encoder.load_state_dict(state_dict)


def count_params(model):
    return sum([np.prod(p.size()) for p in model.parameters() if p.requires_grad])
