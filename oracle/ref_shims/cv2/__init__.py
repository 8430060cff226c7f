"""Stub so util/util.py and data/util.py import; no function is ever called."""
IMREAD_UNCHANGED = -1
def imread(*a, **k): raise NotImplementedError("shim")
def imwrite(*a, **k): raise NotImplementedError("shim")
