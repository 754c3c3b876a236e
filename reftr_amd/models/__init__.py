"""Factory with the reference's name and dispatch (models/__init__.py:4-10)."""
from .reftr_transformer import build_reftr as build_transformer_based_reftr
from .reftr_transformer import build_reftr_seg


def build_reftr(args):
    if getattr(args, "masks", False):
        return build_reftr_seg(args)
    return build_transformer_based_reftr(args)
