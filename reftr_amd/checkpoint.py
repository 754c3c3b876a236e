"""Checkpoint save / resume with the reference's file format (main_vg.py:306-349, 372-384): a dict
{'model', 'optimizer', 'lr_scheduler', 'epoch', 'args', 'best_val_acc'} whose 'model' entry has the reference's
state_dict keys and whose 'optimizer' entry is a torch.optim.AdamW state_dict in the reference's parameter order — files
written by the reference resume here and files written here resume under the reference."""
import copy

import torch

from .util import misc as utils


def save_checkpoint(path, model, optimizer, lr_scheduler, epoch, args=None, best_val_acc=0.0):
    """main_vg.py:377-384 (utils.save_on_master)."""
    inner = getattr(model, "module", model)
    utils.save_on_master({
        "model": inner.state_dict(), "optimizer": optimizer.state_dict(), "lr_scheduler": lr_scheduler.state_dict(),
        "epoch": epoch, "args": args, "best_val_acc": best_val_acc}, path)


def load_checkpoint(checkpoint, model, optimizer=None, lr_scheduler=None, args=None, steps_per_epoch=None):
    """main_vg.py:306-337.  `checkpoint` is a path or an already loaded dict.  Returns (start_epoch, best_val_acc,
    missing_keys, unexpected_keys)."""
    if not isinstance(checkpoint, dict):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    inner = getattr(model, "module", model)
    missing, unexpected = inner.load_state_dict(checkpoint["model"], strict=False)
    unexpected = [k for k in unexpected if not (k.endswith("total_params") or k.endswith("total_ops"))]
    start_epoch = 0
    evaluating = bool(getattr(args, "eval", False))
    model_only = bool(getattr(args, "resume_model_only", False))
    if (optimizer is not None and lr_scheduler is not None and not evaluating and not model_only
            and all(k in checkpoint for k in ("optimizer", "lr_scheduler", "epoch"))):
        p_groups = copy.deepcopy([{k: v for k, v in g.items() if k != "params"} for g in optimizer.param_groups])
        optimizer.load_state_dict(checkpoint["optimizer"])
        for pg, pg_old in zip(optimizer.param_groups, p_groups):        # the command line's learning rates win (:322-324)
            pg["lr"] = pg_old["lr"]
            if "initial_lr" in pg_old:
                pg["initial_lr"] = pg_old["initial_lr"]
        lr_scheduler.load_state_dict(checkpoint["lr_scheduler"])
        if args is not None and hasattr(args, "lr_drop") and hasattr(lr_scheduler, "step_size"):     # :327-333
            lr_scheduler.step_size = args.lr_drop
            lr_scheduler.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
        if steps_per_epoch is not None:
            lr_scheduler.step(steps_per_epoch * lr_scheduler.last_epoch)
        start_epoch = checkpoint["epoch"] + 1
    return start_epoch, checkpoint.get("best_val_acc", 0), missing, unexpected
