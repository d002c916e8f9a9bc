import torch


def dict_send_to(data, device, detach=False, as_numpy=False):
    """Move every tensor of a dict to `device` (same contract as the reference's utils.dict_send_to)."""
    out = {}
    for key, t in data.items():
        if isinstance(t, torch.Tensor):
            if detach:
                t = t.detach()
            t = t.to(device)
            if as_numpy:
                t = t.numpy()
        out[key] = t
    return out
