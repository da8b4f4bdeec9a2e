"""t-svgp_amd: MI355X-native natural-gradient E-step of t-SVGP behind the reference's GPflow-style API.

The directory name carries a hyphen (as the project layout prescribes), so import it with
``importlib.import_module("t-svgp_amd")`` or through the alias module ``tsvgp_amd`` at the repository root.
"""
from . import _backend, distributed, training, util
from ._backend import HipExtensionError, build_library
from .base import Parameter, default_float, default_jitter
from .inducing_variables import InducingPoints, SharedIndependentInducingVariables, inducingpoint_wrapper
from .kernels import Matern32, Matern52, SeparateIndependent, SquaredExponential
from .likelihoods import Bernoulli, Gaussian
from .models import base_SVGP, t_SVGP, t_SVGP_white
from .sites import DenseSites, Sites

__all__ = [
    "t_SVGP", "t_SVGP_white", "base_SVGP", "DenseSites", "Sites", "SquaredExponential", "Gaussian", "Bernoulli", "InducingPoints",
    "SeparateIndependent", "SharedIndependentInducingVariables", "Matern32", "Matern52",
    "inducingpoint_wrapper", "Parameter", "default_float", "default_jitter", "HipExtensionError", "build_library",
    "distributed", "util", "training",
]
