"""Operator-level drop-ins for the un-vendored CUDA extensions the reference imports."""
