"""with PYTHONPATH=<package>:<reference checkout> the reference's own drivers import against
this package.  Runs in the build container only (the reference never travels to the GPU box): executes the import
block of /root/reference/train.py (lines 1-21) and eval.py (lines 1-22) in a fresh interpreter, with empty stand-ins
for the third-party modules this image lacks (tensorboard, editdistance, librosa, soundfile, fastdtw -- none of them on
the hot path), and checks which file every hot-path name resolved to."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import PKG

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train.py")), reason="reference checkout not present")

PROBE = textwrap.dedent('''
    import importlib.machinery, os, sys, types
    def stub(name, **attrs):
        m = types.ModuleType(name); m.__spec__ = importlib.machinery.ModuleSpec(name, None); m.__dict__.update(attrs); sys.modules[name] = m; return m
    for n in ("editdistance", "librosa", "librosa.filters", "librosa.effects", "soundfile", "fastdtw"):
        if n not in sys.modules:
            try: __import__(n)
            except Exception: stub(n, fastdtw=None)
    try:
        import torch.utils.tensorboard  # noqa
    except Exception:
        import torch.utils
        tb = stub("torch.utils.tensorboard", SummaryWriter=object); torch.utils.tensorboard = tb
    ref, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    src = "".join(open(ref).readlines()[first - 1:last])
    ns = {"__name__": "dropin_probe"}
    exec(compile(src, ref, "exec"), ns)
    import utils, synthesize, transformer.tacotron, hyperparams, utils.checkpoint, utils.text
    for name, mod in (("utils", utils), ("utils.checkpoint", utils.checkpoint), ("utils.text", utils.text), ("utils.infolog", ns["infolog"]),
                      ("synthesize", synthesize), ("transformer.tacotron", transformer.tacotron), ("hyperparams", hyperparams),
                      ("dataloader", sys.modules["dataloader"])):
        print("MOD", name, os.path.realpath(mod.__file__))
    assert ns["eval_batch"] is synthesize.eval_batch and ns["save_eval_results"] is synthesize.save_eval_results
    assert ns["dict_send_to"] is utils.dict_send_to and ns["tacotron"] is transformer.tacotron
    print("OK")
''')


def _run_probe(script, first, last):
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + REF)
    r = subprocess.run([sys.executable, "-c", PROBE, os.path.join(REF, script), str(first), str(last)], env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, "import block of %s failed:\n%s\n%s" % (script, r.stdout, r.stderr[-3000:])
    return dict(line.split()[1:3] for line in r.stdout.splitlines() if line.startswith("MOD "))


@needs_ref
@pytest.mark.parametrize("script,first,last", [("train.py", 1, 21), ("eval.py", 1, 22)])
def test_reference_driver_import_block(script, first, last):
    where = _run_probe(script, first, last)
    pkg, ref = os.path.realpath(PKG), os.path.realpath(REF)
    for name in ("utils", "utils.checkpoint", "utils.text", "synthesize", "transformer.tacotron", "hyperparams"):
        assert where[name].startswith(pkg + os.sep), "%s resolved to %s, not to this package" % (name, where[name])
    for name in ("utils.infolog", "dataloader"):           # out-of-scope glue keeps coming from the reference checkout
        assert where[name].startswith(ref + os.sep), "%s resolved to %s" % (name, where[name])


def test_save_eval_results_writes_trimmed_mels(tmp_path):
    """<name>.npy = mel_aft[i][:generated_lengths[i]] (reference synthesize.py:79-81); a bad sample is logged, not raised."""
    import synthesize
    rng = np.random.default_rng(0)
    mel_aft = rng.standard_normal((3, 12, 80)).astype(np.float32)
    out = {"names": ["a", "b", "c"], "mel_pre": mel_aft, "mel_aft": mel_aft, "alignments": {"encdec": [rng.random((3, 2, 5, 12))]},
           "input_lengths": [5, 4, 3], "generated_lengths": [12, 7, 1]}
    synthesize.save_eval_results(**out, output_dir=str(tmp_path / "o"), n_plot_alignment=0)
    for i, n in enumerate(out["names"]):
        got = np.load(tmp_path / "o" / ("%s.npy" % n))
        assert got.shape == (out["generated_lengths"][i], 80) and np.array_equal(got, mel_aft[i][:out["generated_lengths"][i]])
    synthesize.save_eval_results(["x"], mel_aft[:1], mel_aft[:1], {"encdec": []}, [5], [None], str(tmp_path / "o"))      # logged, not raised
    assert not (tmp_path / "o" / "x.npy").exists()
