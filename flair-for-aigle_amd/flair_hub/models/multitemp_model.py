"""Sentinel time-series encoder under the reference's module path (flair_hub/models/multitemp_model.py: ``UTAE``).

The arithmetic lives in flairhip/utae.py (HipUTAE: libflairhip kernels, evaluation-mode forward, the reference's
parameter names); this module keeps the import path and the constructor signature the reference's
FLAIR_HUB_Model uses (flair_model.py:117-134)."""
from flairhip.utae import HipUTAE as UTAE  # noqa: F401

__all__ = ["UTAE"]
