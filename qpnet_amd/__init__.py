"""qpnet_amd -- MI355X-native QPNet vocoder hot path (HIP kernels behind a C ABI).

Public surface mirrors the reference module ``src/nets/qpnet.py`` of bigpon/QPNet:
``encode_mu_law``, ``decode_mu_law``, ``initialize`` and ``QPNet``.  The drop-in shim
that the reference's ``bin/qpnet_*.py`` scripts pick up is ``qpnet_amd/dropin/qpnet.py``
(see INTEGRATION.md).
"""
from .config import QPNetConfig  # noqa: F401

__all__ = ["QPNetConfig", "QPNet", "encode_mu_law", "decode_mu_law", "initialize"]


def __getattr__(name):
    # lazy: importing the package must not require torch/the HIP library (CPU-only tooling)
    if name in ("QPNet", "encode_mu_law", "decode_mu_law", "initialize"):
        from . import qpnet as _q
        return getattr(_q, name)
    raise AttributeError(name)
