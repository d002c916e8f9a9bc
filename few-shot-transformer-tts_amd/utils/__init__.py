"""Small host utilities of the drop-in surface (the reference's `utils` package keeps its other modules -- audio, infolog --
when this package is placed in front of a reference checkout)."""
import torch


def dict_send_to(data, device, detach=False, as_numpy=False):
    """Batch / result dicts mix tensors with plain Python values (`names`): move the tensors to `device`, optionally
    detached and / or as NumPy arrays, and pass everything else through (utils/__init__.py:3-14 of the reference)."""
    def move(v):
        if not isinstance(v, torch.Tensor):
            return v
        v = (v.detach() if detach else v).to(device)
        return v.numpy() if as_numpy else v
    return {k: move(v) for k, v in data.items()}
