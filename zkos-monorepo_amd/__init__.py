"""shielder-hip: MI355X-native Halo2/KZG proving backend for Shielder (host-side Python mirror).

The product is `libshielder_hip.so` (hand-written HIP for gfx950 behind the C ABI of
include/shielder_hip.h).  This package is the thin ctypes mirror used by tests, bench.py and
__graft_entry__.py; PyTorch is used only for device memory, streams and torch.distributed.
It never imports `oracle/` and has no CPU fallback: every compute call fails loudly when the
HIP library or a GPU is missing.
"""
from .binding import *  # noqa: F401,F403
