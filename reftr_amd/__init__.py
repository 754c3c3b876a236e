"""reftr_amd — MI355X-native (gfx950) implementation of the RefTR training hot path.

Public surface mirrors the reference (ubc-vision/RefTR) Python protocol; see INTEGRATION.md.
"""
__version__ = "0.1.0"
