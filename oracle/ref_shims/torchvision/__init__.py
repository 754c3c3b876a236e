"""Minimal stand-in for the `torchvision` package, used ONLY by oracle/gen_golden.py inside the build
container to import /root/reference (torchvision is not installed here and the reference does not vendor
it — SURVEY.md §8c).  It restates, from the public definitions, the handful of symbols the reference's
hot path touches.  It never travels into the product path and nothing under reftr_amd/ imports it.

Consequence (stated in DESIGN.md): the ResNet arithmetic is *defined* by this restatement of ResNet v1.5,
not by a pinned torchvision build — backbone parity is "pinned to the public architecture", everything
else in the golden vectors is produced by the reference's own code.
"""
__version__ = "0.9.0"  # util/misc.py:460 parses this to pick torchvision.ops.misc.interpolate

from . import models, ops  # noqa: E402,F401
