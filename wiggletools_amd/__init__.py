"""wiggletools_amd -- MI355X-native multiplexer / reducer engine for WiggleTools.

Only the hot path (Multiplexer -> reducers / set comparisons) lives here; see
DESIGN.md.  The compute path is the HIP library wiggletools_amd/csrc/
libwiggletools_amd.so behind the C ABI of include/wiggletools_amd.h; importing
`wiggletools_amd.engine` without it raises ImportError (no CPU fallback).
"""
from .runlists import RunLists, synth  # noqa: F401

__version__ = "0.1.0"
