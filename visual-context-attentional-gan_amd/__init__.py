"""MI355X-native VCA-GAN hot path: HIP kernels (csrc/, C ABI in include/vca_hip.h) behind PyTorch modules
that mirror the reference's src/models classes.  Import as `vcagan_amd` (see vcagan_amd.py at the repo root)."""
from . import _lib  # noqa: F401

__all__ = ['_lib']
