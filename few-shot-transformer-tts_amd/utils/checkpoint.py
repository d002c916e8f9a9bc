"""Checkpoint interchange with the reference's on-disk format (utils/checkpoint.py:8-58).

A checkpoint is one `torch.save`d dict with up to four entries -- 'model' (the Tacotron state_dict, never with a
DDP `module.` prefix), 'optim' (torch.optim.Adam.state_dict() layout), 'sched' (LambdaLR.state_dict() layout) and
'step' -- stored as `<model_dir>/model.ckpt-<step>`.  Everything that quacks like the torch objects works on both
sides: a torch optimizer / scheduler pair of a train.py-style loop, or the fused `HipTrainer` (pass it as `optim`
and `trainer.sched` as `sched`), which reads and writes the very same layouts -- so checkpoints published for the
reference (README.md:253-269) resume under the fused trainer, and checkpoints written here resume under train.py.
"""
import logging
import os
import re

import torch

_CKPT_RE = re.compile(r"^model\.ckpt-(\d+)$")
_PARTS = ("model", "optim", "sched")


def find_ckpt(base_dir):
    """Path of the highest-numbered `model.ckpt-<step>` under base_dir (None if there is none or step <= 0)."""
    found = {}
    if os.path.isdir(base_dir):
        for entry in os.listdir(base_dir):
            m = _CKPT_RE.match(entry)
            if m and int(m.group(1)) > 0:
                found[int(m.group(1))] = os.path.join(base_dir, entry)
    return found[max(found)] if found else None


def _bare(model):
    """The module whose state_dict keys carry no wrapper prefix (DistributedDataParallel / DataParallel -> .module)."""
    return getattr(model, "module", model)


def _strip_prefix(sd, prefix="module."):
    """Tolerate state dicts that were saved from a wrapped model (keys 'module.xyz'): the reference never writes
    them, but checkpoints produced by other tools around it do."""
    if sd and all(k.startswith(prefix) for k in sd):
        return type(sd)((k[len(prefix):], v) for k, v in sd.items())
    return sd


def save_model(model_dir, model=None, optim=None, sched=None, step=None):
    """Write {'model', 'optim', 'sched', 'step'} (whichever are given).  With a step the file is
    <model_dir>/model.ckpt-<step>; without one `model_dir` itself is the file name (reference behaviour)."""
    objs = {"model": _bare(model) if model is not None else None, "optim": optim, "sched": sched}
    if hasattr(optim, "sync"):        # fused HIP trainer: an overlapped optimizer step may still be writing the parameters
        optim.sync()
    payload = {k: objs[k].state_dict() for k in _PARTS if objs[k] is not None}
    target = model_dir
    if step:
        payload["step"] = step
        target = os.path.join(model_dir, "model.ckpt-%d" % step)
    torch.save(payload, target)
    return target


def load_model(model_path, model=None, optim=None, sched=None, map_location={}):
    """Restore whichever of model / optim / sched are both in the file and passed in; returns the step (taken from
    the scheduler when the file has none)."""
    payload = torch.load(model_path, map_location=map_location)
    if hasattr(optim, "sync"):
        optim.sync()
    if model is not None and "model" in payload:
        _bare(model).load_state_dict(_strip_prefix(payload["model"]))
    if optim is not None and "optim" in payload:
        optim.load_state_dict(payload["optim"])
    step = payload.get("step")
    if sched is not None and "sched" in payload:
        sched.load_state_dict(payload["sched"])
        if not step:
            step = sched.last_epoch
        elif step != sched.last_epoch:
            logging.warning("Step=%d, while in sched step=%d" % (step, sched.last_epoch))
    return step
