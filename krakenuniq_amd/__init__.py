"""krakenuniq_amd -- MI355X-native KrakenUniq classify hot path.

Layout:
  csrc/                 HIP kernels + the C ABI (include/krakenuniq_amd.h) + the
                        `classify`-compatible CLI, built into libkrakenuniq_amd.so
  capi.py               ctypes binding of the C ABI (plumbing for tests / bench)
  synth.py              deterministic synthetic DB / taxonomy / read generator
"""
__all__ = ["capi", "synth"]
