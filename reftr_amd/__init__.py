"""reftr_amd — MI355X-native (gfx950) implementation of the RefTR training hot path.

Public surface mirrors the reference (ubc-vision/RefTR) Python protocol; see INTEGRATION.md:
    from reftr_amd import build_reftr                      # models/__init__.py:4
    from reftr_amd.engine_vg import train_one_epoch, evaluate
    from reftr_amd.optim import build_optimizer
    from reftr_amd.parallel import DistributedDataParallel
"""
__version__ = "0.1.0"

from .models import build_reftr  # noqa: E402,F401
