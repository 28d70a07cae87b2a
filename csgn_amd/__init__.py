"""csgn_amd -- MI355X-native implementation of the certFHE/CSGN ciphertext-arithmetic hot path.

The product is two native libraries built in-tree for gfx950:

  csgn_amd/lib/libcsgn_hip.so   hand-written HIP kernels behind the C ABI of include/csgn_hip.h
  csgn_amd/lib/libcertFHE.so    the drop-in certFHE::{Context,SecretKey,Plaintext,Ciphertext,...}
                                C++ classes (include/certfhe/) implemented on that ABI

This Python package is plumbing for tests and bench.py only: a ctypes binding of the C ABI
(`csgn_amd.capi`) and torch-backed device buffers / torch.distributed sharding helpers
(`csgn_amd.batch`).  It never computes ciphertext arithmetic on the CPU: if the HIP library
or the GPU is missing, calls raise.
"""
from .capi import CsgnError, load_library, lib_path  # noqa: F401

__all__ = ["CsgnError", "load_library", "lib_path"]
