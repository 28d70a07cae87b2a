"""ctypes binding of include/csgn_hip.h (libcsgn_hip.so).

Every wrapper returns nothing and raises CsgnError(status, message) on failure -- there is no
fallback path.  Device pointers are plain ints (e.g. torch.Tensor.data_ptr()); `stream` is a
hipStream_t handle as int (torch.cuda.current_stream().cuda_stream) or 0.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(PKG, "lib", "libcsgn_hip.so")

CSGN_OK = 0
CSGN_ERR_INVALID = -1
CSGN_ERR_UNSUPPORTED = -2
CSGN_ERR_NO_DEVICE = -3
CSGN_ERR_HIP = -4

u64 = C.c_uint64
vp = C.c_void_p

# name -> (restype, argtypes).  Must list EVERY symbol declared in include/csgn_hip.h
# (tests/test_capi_symbols.py cross-checks this table against the header).
SIGNATURES = {
    "csgn_abi_version": (C.c_int, []),
    "csgn_last_error": (C.c_char_p, []),
    "csgn_init": (C.c_int, [C.c_int]),
    "csgn_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "csgn_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(u64)]),
    "csgn_malloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "csgn_free": (C.c_int, [vp]),
    "csgn_host_alloc": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.c_size_t]),
    "csgn_host_free": (C.c_int, [vp]),
    "csgn_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "csgn_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "csgn_memcpy_d2d": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "csgn_memset": (C.c_int, [vp, C.c_int, C.c_size_t, vp]),
    "csgn_stream_create": (C.c_int, [C.POINTER(vp)]),
    "csgn_stream_destroy": (C.c_int, [vp]),
    "csgn_stream_sync": (C.c_int, [vp]),
    "csgn_event_create": (C.c_int, [C.POINTER(vp)]),
    "csgn_event_destroy": (C.c_int, [vp]),
    "csgn_event_record": (C.c_int, [vp, vp]),
    "csgn_event_sync": (C.c_int, [vp]),
    "csgn_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "csgn_default_len": (u64, [u64]),
    "csgn_context_s": (u64, [u64, u64]),
    "csgn_mul_len": (u64, [u64, u64, u64]),
    "csgn_bitlen_canonical": (C.c_int, [u64, u64, vp]),
    "csgn_key_mask": (C.c_int, [u64, vp, u64, vp]),
    "csgn_mul_uniform": (C.c_int, [u64, u64, u64, u64, vp, vp, vp, u64, vp]),
    "csgn_mul_ragged_plan": (C.c_int, [u64, vp, vp, vp, C.POINTER(u64 * 4), vp]),
    "csgn_mul_ragged": (C.c_int, [u64, u64, vp, vp, vp, vp, vp, vp, u64, u64, u64, vp]),
    "csgn_mul_plan_create": (C.c_int, [C.POINTER(vp)]),
    "csgn_mul_plan_destroy": (None, [vp]),
    "csgn_mul_plan_ragged": (C.c_int, [vp, u64, vp, vp, vp, C.POINTER(u64 * 4), vp]),
    "csgn_mul_planned": (C.c_int, [vp, u64, vp, vp, vp, vp]),
    "csgn_mul_plan_validate": (C.c_int, [vp, vp]),
    "csgn_mul_plan_trust": (C.c_int, [vp, C.c_int]),
    "csgn_mul_ragged_async_plan_words": (u64, [u64]),
    "csgn_mul_ragged_async": (C.c_int, [u64, u64, vp, vp, vp, vp, vp, vp, u64, vp, vp]),
    "csgn_mul_ragged_async_result": (C.c_int, [vp, C.POINTER(u64 * 5), vp]),
    "csgn_add_uniform": (C.c_int, [u64, u64, u64, u64, vp, vp, vp, vp]),
    "csgn_add_ragged": (C.c_int, [u64, u64, vp, vp, vp, vp, vp, vp, u64, vp]),
    "csgn_small_ops": (C.c_int, [u64, u64, vp, vp]),
    "csgn_add_ragged_bounded": (C.c_int, [u64, u64, u64, u64, vp, vp, vp, vp, vp, vp, u64, vp]),
    "csgn_decrypt_scratch_bytes": (C.c_size_t, [u64, u64]),
    "csgn_decrypt_uniform": (C.c_int, [u64, u64, u64, vp, vp, vp, vp, vp]),
    "csgn_decrypt_ragged": (C.c_int, [u64, u64, u64, vp, vp, vp, vp, vp, vp]),
    "csgn_decrypt_ragged_bounded": (C.c_int, [u64, u64, u64, u64, vp, vp, vp, vp, vp, vp]),
    "csgn_decrypt_combined_scratch_bytes": (C.c_size_t, [u64, u64, u64]),
    "csgn_decrypt_product_uniform": (C.c_int, [u64, u64, u64, u64, vp, vp, vp, vp, vp, vp]),
    "csgn_decrypt_sum_uniform": (C.c_int, [u64, u64, u64, u64, vp, vp, vp, vp, vp, vp]),
    "csgn_compact_scratch_bytes": (C.c_size_t, [u64, u64, u64]),
    "csgn_compact_ragged": (C.c_int, [u64, u64, u64, u64, vp, vp, vp, vp, vp, vp]),
    "csgn_encrypt_explicit": (C.c_int, [u64, u64, u64, vp, vp, vp, vp, vp, vp, vp]),
    "csgn_rng_from_os": (C.c_int, [vp, C.c_uint32]),
    "csgn_rng_from_seed": (C.c_int, [vp, u64, C.c_uint32]),
    "csgn_encrypt_keyed_layout": (C.c_int, [u64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "csgn_encrypt_keyed": (C.c_int, [u64, u64, u64, u64, vp, vp, vp, vp, vp, vp]),
    "csgn_encrypt_mul_keyed": (C.c_int, [u64, u64, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "csgn_encrypt_device_rng": (C.c_int, [u64, u64, u64, vp, vp, vp, u64, vp, vp]),
    "csgn_permute_uniform": (C.c_int, [u64, u64, u64, C.c_int, vp, vp, vp, vp]),
    "csgn_bitlen_scratch_bytes": (C.c_size_t, [u64]),
    "csgn_decrypt_bitlen": (C.c_int, [u64, u64, u64, vp, vp, vp, vp, vp, vp]),
    "csgn_permute_bitlen": (C.c_int, [u64, u64, vp, vp, vp, vp, vp, vp]),
    "csgn_synth_fill": (C.c_int, [u64, u64, u64, u64, vp, vp]),
    "csgn_digest": (C.c_int, [vp, u64, u64, vp, vp]),
    "csgn_circuit_create": (C.c_int, [u64, u64, C.POINTER(vp)]),
    "csgn_circuit_destroy": (None, [vp]),
    "csgn_circuit_input": (C.c_int, [vp, u64, C.POINTER(C.c_uint32)]),
    "csgn_circuit_input_ragged": (C.c_int, [vp, vp, C.POINTER(C.c_uint32)]),
    "csgn_circuit_add": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "csgn_circuit_mul": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "csgn_circuit_decrypt": (C.c_int, [vp, C.c_uint32, vp, C.POINTER(C.c_uint32)]),
    "csgn_circuit_compact": (C.c_int, [vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "csgn_circuit_permute": (C.c_int, [vp, C.c_uint32, vp, C.POINTER(C.c_uint32)]),
    "csgn_circuit_encrypt": (C.c_int, [vp, u64, vp, vp, vp, vp, u64, C.POINTER(C.c_uint32)]),
    "csgn_circuit_encrypt_mul": (C.c_int, [vp, u64, vp, vp, vp, vp, vp, vp, u64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "csgn_circuit_epoch": (u64, [vp]),
    "csgn_circuit_node_key": (C.c_int, [vp, vp]),
    "csgn_circuit_optimize": (C.c_int, [vp, C.c_uint32]),
    "csgn_circuit_output": (C.c_int, [vp, C.c_uint32]),
    "csgn_circuit_block_bytes": (u64, [vp]),
    "csgn_circuit_stats": (C.c_int, [vp, vp]),
    "csgn_circuit_plan_json": (C.c_int, [vp, vp, C.c_size_t]),
    "csgn_circuit_build": (C.c_int, [vp]),
    "csgn_circuit_value": (vp, [vp, C.c_uint32]),
    "csgn_circuit_value_terms": (u64, [vp, C.c_uint32]),
    "csgn_circuit_value_total_terms": (u64, [vp, C.c_uint32]),
    "csgn_circuit_value_offsets": (vp, [vp, C.c_uint32]),
    "csgn_circuit_bits": (vp, [vp, C.c_uint32]),
    "csgn_circuit_run": (C.c_int, [vp, vp]),
    "csgn_mul_uniform_kernel": (C.c_char_p, [u64, u64, u64, u64]),
    "csgn_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "csgn_get_tuning": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "csgn_reset_tuning": (None, []),
    "csgn_tuning_name": (C.c_char_p, [C.c_int]),
    "csgn_debug_fastdiv": (C.c_uint32, [C.c_uint32, C.c_uint32]),
}


class CsgnRng(C.Structure):
    """csgn_rng of include/csgn_hip.h: ChaCha key / nonce / rounds of the keyed device generator."""
    _fields_ = [("key", C.c_uint32 * 8), ("nonce", C.c_uint64), ("rounds", C.c_uint32), ("reserved", C.c_uint32)]


class CsgnError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"csgn status {status}: {message}")
        self.status = status
        self.message = message


def lib_path() -> str:
    return _LIB_PATH


_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libcsgn_hip.so (once).  Raises if it has not been built -- run
    `python -m csgn_amd.build` (or __graft_entry__.build()) first."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("CSGN_HIP_LIB") or _LIB_PATH   # env override: A/B of library builds
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} is missing: the HIP library has not been built (python -m csgn_amd.build). "
            "csgn_amd has no CPU fallback.")
    try:
        # torch bundles its own libamdhip64.so.7; importing it first makes this library bind
        # to that same runtime instance, so torch streams/tensors can be handed across.
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != CSGN_OK:
        msg = load_library().csgn_last_error()
        raise CsgnError(rc, msg.decode("utf-8", "replace") if msg else "")


# -- tuning knobs (csgn_set_tuning; csgn_amd/csrc/csgn_tuning.h) ---------------------------------
def _knob(name: str) -> bytes:
    """'mul_flat' or the environment-style 'CSGN_MUL_FLAT'."""
    n = name[5:] if name.upper().startswith("CSGN_") else name
    return n.lower().encode()


def set_tuning(name: str, value) -> None:
    check(load_library().csgn_set_tuning(_knob(name), int(value)))


def get_tuning(name: str) -> int:
    v = C.c_int(0)
    check(load_library().csgn_get_tuning(_knob(name), C.byref(v)))
    return v.value


def reset_tuning() -> None:
    load_library().csgn_reset_tuning()


def tuning_names():
    lib, out, i = load_library(), [], 0
    while True:
        n = lib.csgn_tuning_name(i)
        if not n:
            return out
        out.append(n.decode())
        i += 1


# -- libcsgn_shard.so (include/csgn_shard.h): partition + RCCL all-gather of term counts ---------
_SHARD_LIB_PATH = os.path.join(PKG, "lib", "libcsgn_shard.so")
CSGN_COMM_ID_BYTES = 128
CSGN_STREAM_OF_COMM = C.c_void_p(-1).value        # (void *)-1: the communicator's own stream
CSGN_ERR_TIMEOUT = -5
CSGN_COMM_STRICT = 0
CSGN_COMM_ALLOW_MINOR_SKEW = 1
CSGN_COMM_OPT_FORCE_GROUPED_BROADCAST = 1

SHARD_SIGNATURES = {
    "csgn_shard_last_error": (C.c_char_p, []),
    "csgn_shard_range": (C.c_int, [u64, C.c_int, C.c_int, C.POINTER(u64), C.POINTER(u64)]),
    "csgn_shard_owner": (C.c_int, [u64, u64, C.c_int]),
    "csgn_comm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "csgn_comm_rccl_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "csgn_shard_gather_plan": (C.c_int, [u64, C.c_int, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_int)]),
    "csgn_comm_init_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]),
    "csgn_comm_init_all_ex": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_uint, C.POINTER(vp)]),
    "csgn_comm_unique_id": (C.c_int, [C.c_char_p]),
    "csgn_comm_init_rank": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "csgn_comm_init_rank_ex": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(vp)]),
    "csgn_comm_destroy": (C.c_int, [vp]),
    "csgn_comm_abort": (C.c_int, [vp]),
    "csgn_comm_check": (C.c_int, [vp]),
    "csgn_comm_set_timeout_ms": (C.c_int, [vp, u64]),
    "csgn_comm_set_option": (C.c_int, [vp, C.c_int, C.c_int]),
    "csgn_comm_rank": (C.c_int, [vp]),
    "csgn_comm_world": (C.c_int, [vp]),
    "csgn_comm_device": (C.c_int, [vp]),
    "csgn_comm_stream": (vp, [vp]),
    "csgn_comm_gather_counts": (C.c_int, [vp, vp, u64, vp, vp]),
    "csgn_comm_gather_bytes": (C.c_int, [vp, vp, u64, vp, vp]),
    "csgn_comm_barrier": (C.c_int, [vp, vp]),
    "csgn_shard_product_counts": (C.c_int, [u64, vp, vp, u64, u64, vp, vp]),
}

_shard_lib: Optional[C.CDLL] = None


def load_shard_library() -> C.CDLL:
    """Load libcsgn_shard.so (once).  It pulls in librccl; import torch first so that both bind to
    the one RCCL / HIP runtime torch ships."""
    global _shard_lib
    if _shard_lib is not None:
        return _shard_lib
    if not os.path.exists(_SHARD_LIB_PATH):
        raise FileNotFoundError(f"{_SHARD_LIB_PATH} is missing (python -m csgn_amd.build)")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(_SHARD_LIB_PATH)
    for name, (res, args) in SHARD_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _shard_lib = lib
    return lib


def rccl_info():
    """(runtime version code, header version code, path of the librccl the process bound)."""
    lib = load_shard_library()
    rt, hd = C.c_int(0), C.c_int(0)
    path = C.create_string_buffer(1024)
    check_shard(lib.csgn_comm_rccl_info(C.byref(rt), C.byref(hd), path, len(path)))
    return rt.value, hd.value, path.value.decode()


def rccl_version_text(code: int) -> str:
    return "%d.%d.%d" % (code // 10000, (code // 100) % 100, code % 100)


def check_shard(rc: int) -> None:
    if rc != CSGN_OK:
        msg = load_shard_library().csgn_shard_last_error()
        raise CsgnError(rc, msg.decode("utf-8", "replace") if msg else "")
