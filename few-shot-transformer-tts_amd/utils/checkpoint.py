"""Checkpoint interchange with the reference format (utils/checkpoint.py:8-58):
torch.save({'model', 'optim', 'sched', 'step'}) to <model_dir>/model.ckpt-<step>; a DDP/DataParallel
`.module` wrapper is unwrapped on both save and load so published checkpoints round-trip."""
import glob
import logging
import os

import torch


def find_ckpt(base_dir):
    best, best_step = None, 0
    for f in glob.iglob(os.path.join(base_dir, "model.ckpt-*")):
        step = int(f.split("-")[-1])
        if step > best_step:
            best, best_step = f, step
    return best


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def save_model(model_dir, model=None, optim=None, sched=None, step=None):
    state = {}
    if model:
        state["model"] = _unwrap(model).state_dict()
    if optim:
        state["optim"] = optim.state_dict()
    if sched:
        state["sched"] = sched.state_dict()
    if step:
        state["step"] = step
        model_dir = os.path.join(model_dir, "model.ckpt-%d" % step)
    torch.save(state, model_dir)


def load_model(model_path, model=None, optim=None, sched=None, map_location={}):
    state = torch.load(model_path, map_location=map_location)
    if "model" in state and model:
        _unwrap(model).load_state_dict(state["model"])
    if "optim" in state and optim:
        optim.load_state_dict(state["optim"])
    step = state.get("step")
    if "sched" in state and sched:
        sched.load_state_dict(state["sched"])
        if step:
            if step != sched.last_epoch:
                logging.warning("Step=%d, while in sched step=%d" % (step, sched.last_epoch))
        else:
            step = sched.last_epoch
    return step
