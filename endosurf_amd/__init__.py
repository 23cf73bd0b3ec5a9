"""endosurf_amd — MI355X-native (gfx950 HIP) implementation of EndoSurf's per-ray volume-rendering hot path."""
from .renderer import EndoSurfRenderer  # noqa: F401

__all__ = ["EndoSurfRenderer"]
