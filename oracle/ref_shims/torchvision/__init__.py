"""Shim: only what model/GPEMSR.py:11 and model/VGG.py:3 import."""
__version__ = "0.0.shim"
from . import models, ops, utils  # noqa: F401
