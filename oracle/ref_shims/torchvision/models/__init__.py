from . import vgg  # noqa: F401
from .vgg import vgg19  # noqa: F401
