"""Import alias: ``import tsvgp_amd`` -> the package in the directory ``t-svgp_amd/`` (a hyphen is not importable)."""
import importlib
import sys

_pkg = importlib.import_module("t-svgp_amd")
for _name, _mod in list(sys.modules.items()):
    if _name == "t-svgp_amd" or _name.startswith("t-svgp_amd."):
        sys.modules[_name.replace("t-svgp_amd", "tsvgp_amd", 1)] = _mod
sys.modules[__name__] = _pkg
