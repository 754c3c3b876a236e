"""Factory with the reference's name and dispatch (models/__init__.py:4-10)."""
from .reftr_transformer import build_reftr as build_transformer_based_reftr


def build_reftr(args):
    if getattr(args, "masks", False):
        raise NotImplementedError("RefTRSeg (--masks) is SURVEY.md §8 row a17/a18: not built yet in this round")
    return build_transformer_based_reftr(args)
