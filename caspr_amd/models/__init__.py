from .caspr import CaSPR  # noqa: F401
