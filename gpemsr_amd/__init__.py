"""gpemsr_amd -- MI355X-native (gfx950) implementation of the GPEMSR stage-3 super-resolution
forward behind the reference's module / option-file / CLI interface.  See DESIGN.md."""
__all__ = ["GPEMSR"]


def __getattr__(name):
    if name == "GPEMSR":
        from .model import GPEMSR
        return GPEMSR
    raise AttributeError(name)
