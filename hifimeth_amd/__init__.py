"""hifimeth_amd -- MI355X (gfx950) engine for the `hifimeth call` hot path.

Only what the path needs lives here: csrc/ (HIP kernels + the C ABI), the ctypes loader, the
host-side mirror of the reference's feature-generator / batcher interface, the model-file reader
and a synthetic-read generator for tests and benchmarks.
"""
from .caller import CALL_DTYPE, CHG, CHH, CPG, CTX_NAMES, HifimethError, MethylationCaller, parse_contexts  # noqa: F401
