from . import _utils  # noqa: F401
from .resnet import resnet50, resnet101  # noqa: F401
