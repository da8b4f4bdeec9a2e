"""Model classes (mirror of reference src/models)."""
from .tsvgp import base_SVGP, t_SVGP
from .tsvgp_white import t_SVGP_white

__all__ = ["base_SVGP", "t_SVGP", "t_SVGP_white"]
