"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the Transformer-TTS hot path.

This package is a plain fp32 PyTorch-CPU / NumPy restatement of the algorithm in
mutiann/few-shot-transformer-tts (transformer/{attention,modules,tacotron,common}.py
and synthesize.py:eval_batch).  It is the *checker* for the HIP path, never the thing
that is shipped or measured:

  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it;
  * the product package (few-shot-transformer-tts_amd/) never imports it and has no CPU
    fallback -- it raises if the HIP library is missing.

Pinning: the reference holds no tests or golden vectors of its own, so
the oracle is pinned against outputs of the reference itself, generated in the build
container by tests/golden/make_goldens.py (which imports /root/reference) and committed
as data under tests/golden/*.npz.  tests/test_oracle_vs_golden.py checks every fixture.
"""
from .config import default_config, make_config, TINY, TINY96  # noqa: F401
from . import b2s_oracle, synth  # noqa: F401
