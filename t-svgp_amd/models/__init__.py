"""Model classes (mirror of reference src/models)."""
from .tsvgp import base_SVGP, t_SVGP

__all__ = ["base_SVGP", "t_SVGP"]
