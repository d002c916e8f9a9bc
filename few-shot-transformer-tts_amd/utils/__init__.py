"""Small host utilities of the drop-in surface.

Placed in front of a reference checkout (PYTHONPATH=<this package>:<reference>) this package supplies `dict_send_to`,
`utils.checkpoint`, `utils.hparams` and `utils.text`; the reference's other `utils` modules (infolog, audio, transcribe:
logging / plotting / vocoder / cloud-STT glue, out of scope here) stay importable because the package path is extended
with every other `utils` directory on sys.path (tests/test_dropin_imports.py executes the import blocks of the
reference's train.py and eval.py against this layout)."""
import pkgutil

import torch

__path__ = pkgutil.extend_path(__path__, __name__)


def dict_send_to(data, device, detach=False, as_numpy=False):
    """Batch / result dicts mix tensors with plain Python values (`names`): move the tensors to `device`, optionally
    detached and / or as NumPy arrays, and pass everything else through (utils/__init__.py:3-14 of the reference)."""
    def move(v):
        if not isinstance(v, torch.Tensor):
            return v
        v = (v.detach() if detach else v).to(device)
        return v.numpy() if as_numpy else v
    return {k: move(v) for k, v in data.items()}
