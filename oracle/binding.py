"""ctypes bindings for the TEST-ONLY checkers in oracle/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  Nothing under csgn_amd/ does.

  Oracle  -- our CPU restatement (oracle/csgn_oracle.c -> libcsgn_oracle.so)
  Ref     -- the real reference compiled from /root/reference/src (oracle/_ref/libcsgn_ref.so);
             `load_ref()` returns None where that file does not exist.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libcsgn_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libcsgn_ref.so")

u64 = C.c_uint64
i64 = C.c_int64
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


def _p64(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _p32(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(i32p)


def as_u64(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def build_oracle(force: bool = False) -> str:
    """Compile libcsgn_oracle.so (gcc is on every box) if missing or stale."""
    src = os.path.join(HERE, "csgn_oracle.c")
    hdr = os.path.join(HERE, "csgn_oracle.h")
    stale = (not os.path.exists(ORACLE_SO)
             or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def build_ref() -> Optional[str]:
    """Compile oracle/_ref from the reference sources where they exist (dev container)."""
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    return REF_SO if os.path.exists(REF_SO) else None


def build_demos() -> None:
    """The reference's own demo programs linked against the drop-in library (needs
    csgn_amd/lib/libcertFHE.so; dev container only, prebuilt binaries travel to the GPU box)."""
    subprocess.check_call(["make", "-C", HERE, "demos"], stdout=subprocess.DEVNULL)


def glibc_draws(seed: int, count: int) -> np.ndarray:
    """`count` successive rand() results after srand(seed), from this box's libc."""
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    libc.srand.argtypes = [C.c_uint]
    libc.srand(C.c_uint(seed & 0xFFFFFFFF))
    out = np.empty(count, dtype=np.int32)
    for i in range(count):
        out[i] = libc.rand()
    return out


def canonical_bitlen(n_bits: int, terms: int) -> np.ndarray:
    dl = (n_bits + 63) // 64
    rem = n_bits % 64
    one = np.full(dl, 64, dtype=np.uint64)
    if rem:
        one[-1] = rem
    return np.tile(one, terms)


class Oracle:
    def __init__(self, path: Optional[str] = None):
        self.lib = C.CDLL(path or build_oracle())
        L = self.lib
        L.csgn_oracle_default_len.restype = u64
        L.csgn_oracle_default_len.argtypes = [u64]
        L.csgn_oracle_context_s.restype = u64
        L.csgn_oracle_context_s.argtypes = [u64, u64]
        L.csgn_oracle_bitlen.restype = None
        L.csgn_oracle_bitlen.argtypes = [u64, u64, u64p]
        L.csgn_oracle_mul_len.restype = u64
        L.csgn_oracle_mul_len.argtypes = [u64, u64, u64]
        L.csgn_oracle_mul.restype = u64
        L.csgn_oracle_mul.argtypes = [u64, u64p, u64, u64p, u64p, u64, u64p, u64p]
        L.csgn_oracle_mul_reference_cost.restype = u64
        L.csgn_oracle_mul_reference_cost.argtypes = [u64, u64p, u64, u64p, u64]
        L.csgn_oracle_add.restype = u64
        L.csgn_oracle_add.argtypes = [u64p, u64, u64p, u64p, u64, u64p, u64p, u64p]
        L.csgn_oracle_keygen.restype = i64
        L.csgn_oracle_keygen.argtypes = [u64, u64, i32p, u64, u64p]
        L.csgn_oracle_node_key.restype = None
        L.csgn_oracle_node_key.argtypes = [C.c_void_p, u64, C.c_void_p]
        L.csgn_oracle_key_mask.restype = None
        L.csgn_oracle_key_mask.argtypes = [u64, u64p, u64, u64p]
        L.csgn_oracle_encrypt.restype = i64
        L.csgn_oracle_encrypt.argtypes = [u64, u64, u64p, C.c_uint, i32p, u64, u64p]
        L.csgn_oracle_chacha_block.restype = None
        L.csgn_oracle_chacha_block.argtypes = [C.c_void_p, u64, u64, C.c_uint, C.c_void_p]
        L.csgn_oracle_rng_from_seed.restype = None
        L.csgn_oracle_rng_from_seed.argtypes = [u64, C.c_void_p, C.POINTER(u64)]
        L.csgn_oracle_keyed_layout.restype = None
        L.csgn_oracle_keyed_layout.argtypes = [u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.csgn_oracle_encrypt_keyed.restype = None
        L.csgn_oracle_encrypt_keyed.argtypes = [u64, u64, u64p, u64, u64, C.c_void_p, C.c_void_p, u64, C.c_uint, u64p]
        L.csgn_oracle_decrypt.restype = C.c_uint
        L.csgn_oracle_decrypt.argtypes = [u64, u64, u64p, u64p, u64, u64p]
        L.csgn_oracle_decrypt_canonical.restype = C.c_uint
        L.csgn_oracle_decrypt_canonical.argtypes = [u64, u64, u64p, u64p, u64]
        L.csgn_oracle_perm_random.restype = i64
        L.csgn_oracle_perm_random.argtypes = [u64, i32p, u64, u64p]
        L.csgn_oracle_perm_inverse.restype = None
        L.csgn_oracle_perm_inverse.argtypes = [u64p, u64, u64p]
        L.csgn_oracle_perm_compose.restype = C.c_int
        L.csgn_oracle_perm_compose.argtypes = [u64p, u64, u64p, u64, u64p]
        L.csgn_oracle_permute_ciphertext.restype = u64
        L.csgn_oracle_permute_ciphertext.argtypes = [u64, u64p, u64p, u64, u64p, u64p]
        L.csgn_oracle_permute_key.restype = u64
        L.csgn_oracle_permute_key.argtypes = [u64, u64p, u64p, u64, u64p]
        L.csgn_oracle_compact.restype = u64
        L.csgn_oracle_compact.argtypes = [u64, u64p, u64, u64p]
        L.csgn_oracle_synth_word.restype = u64
        L.csgn_oracle_synth_word.argtypes = [u64, u64]
        L.csgn_oracle_synth_fill.restype = None
        L.csgn_oracle_synth_fill.argtypes = [u64, u64, u64, u64, u64p]
        L.csgn_oracle_digest.restype = u64
        L.csgn_oracle_digest.argtypes = [u64p, u64, u64]

    # -- context ------------------------------------------------------------------
    def default_len(self, n_bits: int) -> int:
        return int(self.lib.csgn_oracle_default_len(n_bits))

    def context_s(self, n_bits: int, d: int) -> int:
        return int(self.lib.csgn_oracle_context_s(n_bits, d))

    def bitlen(self, n_bits: int, terms: int) -> np.ndarray:
        out = np.empty(terms * self.default_len(n_bits), dtype=np.uint64)
        self.lib.csgn_oracle_bitlen(n_bits, terms, _p64(out))
        return out

    # -- arithmetic ---------------------------------------------------------------
    def mul(self, n_bits: int, c1, c2, bitlen1=None) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        dl = self.default_len(n_bits)
        c1, c2 = as_u64(c1), as_u64(c2)
        newlen = int(self.lib.csgn_oracle_mul_len(dl, c1.size, c2.size))
        out = np.empty(newlen, dtype=np.uint64)
        blo = None
        if bitlen1 is not None:
            bitlen1 = as_u64(bitlen1)
            blo = np.empty(newlen, dtype=np.uint64)
        got = self.lib.csgn_oracle_mul(dl, _p64(c1), c1.size, _p64(bitlen1), _p64(c2), c2.size,
                                       _p64(out), _p64(blo))
        assert got == newlen
        return out, blo

    def mul_reference_cost(self, n_bits: int, c1, c2) -> int:
        dl = self.default_len(n_bits)
        c1, c2 = as_u64(c1), as_u64(c2)
        return int(self.lib.csgn_oracle_mul_reference_cost(dl, _p64(c1), c1.size, _p64(c2), c2.size))

    def add(self, c1, c2, bitlen1=None, bitlen2=None):
        c1, c2 = as_u64(c1), as_u64(c2)
        out = np.empty(c1.size + c2.size, dtype=np.uint64)
        blo = None
        if bitlen1 is not None:
            bitlen1, bitlen2 = as_u64(bitlen1), as_u64(bitlen2)
            blo = np.empty(out.size, dtype=np.uint64)
        self.lib.csgn_oracle_add(_p64(c1), c1.size, _p64(bitlen1), _p64(c2), c2.size, _p64(bitlen2),
                                 _p64(out), _p64(blo))
        return out, blo

    # -- key / encrypt / decrypt ----------------------------------------------------
    def keygen(self, n_bits: int, d: int, draws: np.ndarray) -> Tuple[np.ndarray, int]:
        key = np.zeros(d, dtype=np.uint64)
        used = int(self.lib.csgn_oracle_keygen(n_bits, d, _p32(draws), draws.size, _p64(key)))
        if used < 0:
            raise ValueError("not enough draws")
        return key, used

    def key_mask(self, n_bits: int, key) -> np.ndarray:
        key = as_u64(key)
        mask = np.zeros(self.default_len(n_bits), dtype=np.uint64)
        self.lib.csgn_oracle_key_mask(n_bits, _p64(key), key.size, _p64(mask))
        return mask

    def encrypt(self, n_bits: int, key, bit: int, draws: np.ndarray) -> Tuple[np.ndarray, int]:
        key = as_u64(key)
        out = np.zeros(self.default_len(n_bits), dtype=np.uint64)
        used = int(self.lib.csgn_oracle_encrypt(n_bits, key.size, _p64(key), bit & 1,
                                                _p32(draws), draws.size, _p64(out)))
        if used < 0:
            raise ValueError("not enough draws")
        return out, used

    def encrypt_seq(self, n_bits: int, key, bits, draws: np.ndarray) -> Tuple[np.ndarray, int]:
        dl = self.default_len(n_bits)
        out = np.zeros(len(bits) * dl, dtype=np.uint64)
        pos = 0
        for i, b in enumerate(bits):
            ct, used = self.encrypt(n_bits, key, int(b), np.ascontiguousarray(draws[pos:]))
            out[i * dl:(i + 1) * dl] = ct
            pos += used
        return out, pos

    # -- keyed device generator (definitions shared with the HIP side) --------------
    def chacha_block(self, key_words, nonce: int, counter: int, rounds: int) -> np.ndarray:
        k = np.ascontiguousarray(np.asarray(key_words, dtype=np.uint32))
        assert k.size == 8
        out = np.zeros(16, dtype=np.uint32)
        self.lib.csgn_oracle_chacha_block(k.ctypes.data, nonce & (2**64 - 1), counter & (2**64 - 1), rounds,
                                          out.ctypes.data)
        return out

    def rng_from_seed(self, seed: int):
        """(key[8] as uint32, nonce) of csgn_rng_from_seed."""
        k = np.zeros(8, dtype=np.uint32)
        nonce = u64(0)
        self.lib.csgn_oracle_rng_from_seed(seed & (2**64 - 1), k.ctypes.data, C.byref(nonce))
        return k, int(nonce.value)

    def node_key(self, rng_key, nonce: int) -> np.ndarray:
        """Key of a circuit encrypt node built from (rng_key, nonce): csgn_circuit_node_key."""
        k = np.ascontiguousarray(np.asarray(rng_key, dtype=np.uint32))
        out = np.zeros(8, dtype=np.uint32)
        self.lib.csgn_oracle_node_key(k.ctypes.data, nonce & (2**64 - 1), out.ctypes.data)
        return out

    def keyed_layout(self, n_bits: int):
        a, b, c = u64(0), u64(0), u64(0)
        self.lib.csgn_oracle_keyed_layout(n_bits, C.byref(a), C.byref(b), C.byref(c))
        return int(a.value), int(b.value), int(c.value)

    def encrypt_keyed(self, n_bits: int, key, plain, rng_key, nonce: int, rounds: int = 8,
                      first_ciphertext: int = 0) -> np.ndarray:
        key = as_u64(key)
        plain = np.ascontiguousarray(np.asarray(plain, dtype=np.uint8))
        rk = np.ascontiguousarray(np.asarray(rng_key, dtype=np.uint32))
        out = np.zeros(plain.size * self.default_len(n_bits), dtype=np.uint64)
        self.lib.csgn_oracle_encrypt_keyed(n_bits, key.size, _p64(key), plain.size, first_ciphertext,
                                           plain.ctypes.data, rk.ctypes.data, nonce & (2**64 - 1), rounds,
                                           _p64(out))
        return out

    def decrypt(self, n_bits: int, key, v, bitlen=None) -> int:
        key, v = as_u64(key), as_u64(v)
        if bitlen is not None:
            bitlen = as_u64(bitlen)
        return int(self.lib.csgn_oracle_decrypt(n_bits, key.size, _p64(key), _p64(v), v.size,
                                                _p64(bitlen)))

    def decrypt_canonical(self, n_bits: int, key, v) -> int:
        key, v = as_u64(key), as_u64(v)
        return int(self.lib.csgn_oracle_decrypt_canonical(n_bits, key.size, _p64(key), _p64(v), v.size))

    # -- permutations -------------------------------------------------------------
    def perm_random(self, size: int, draws: np.ndarray) -> Tuple[np.ndarray, int]:
        perm = np.zeros(size, dtype=np.uint64)
        used = int(self.lib.csgn_oracle_perm_random(size, _p32(draws), draws.size, _p64(perm)))
        if used < 0:
            raise ValueError("not enough draws")
        return perm, used

    def perm_inverse(self, perm) -> np.ndarray:
        perm = as_u64(perm)
        inv = np.zeros(perm.size, dtype=np.uint64)
        self.lib.csgn_oracle_perm_inverse(_p64(perm), perm.size, _p64(inv))
        return inv

    def perm_compose(self, a, b) -> Optional[np.ndarray]:
        a, b = as_u64(a), as_u64(b)
        out = np.zeros(a.size, dtype=np.uint64)
        rc = self.lib.csgn_oracle_perm_compose(_p64(a), a.size, _p64(b), b.size, _p64(out))
        return None if rc != 0 else out

    def permute_ciphertext(self, n_bits: int, perm, v, bitlen=None) -> np.ndarray:
        perm, v = as_u64(perm), as_u64(v)
        if bitlen is not None:
            bitlen = as_u64(bitlen)
        out = np.zeros(self.default_len(n_bits), dtype=np.uint64)
        self.lib.csgn_oracle_permute_ciphertext(n_bits, _p64(perm), _p64(v), v.size, _p64(bitlen),
                                                _p64(out))
        return out

    def permute_key(self, n_bits: int, perm, key) -> np.ndarray:
        perm, key = as_u64(perm), as_u64(key)
        out = np.zeros(key.size, dtype=np.uint64)
        cnt = int(self.lib.csgn_oracle_permute_key(n_bits, _p64(perm), _p64(key), key.size, _p64(out)))
        return out[:cnt]

    # -- extension checker ----------------------------------------------------------
    def compact(self, n_bits: int, v) -> np.ndarray:
        """Mod-2 compaction (extension, not reference behaviour)."""
        v = as_u64(v)
        dl = self.default_len(n_bits)
        out = np.zeros(max(v.size, 1), dtype=np.uint64)
        kept = int(self.lib.csgn_oracle_compact(dl, _p64(v), v.size // dl, _p64(out)))
        return out[: kept * dl].copy()

    # -- harness helpers ------------------------------------------------------------
    def synth(self, seed: int, n_bits: int, first_word: int, n_words: int) -> np.ndarray:
        out = np.empty(n_words, dtype=np.uint64)
        self.lib.csgn_oracle_synth_fill(seed & (2**64 - 1), n_bits, first_word, n_words, _p64(out))
        return out

    def digest(self, w, first_index: int = 0) -> int:
        w = as_u64(w)
        return int(self.lib.csgn_oracle_digest(_p64(w), w.size, first_index))


class Ref:
    """The genuine reference, driven through oracle/ref_driver.cpp."""

    def __init__(self, path: str = REF_SO):
        self.lib = C.CDLL(path)
        L = self.lib
        L.ref_default_len.restype = u64
        L.ref_default_len.argtypes = [u64, u64]
        L.ref_context_s.restype = u64
        L.ref_context_s.argtypes = [u64, u64]
        L.ref_keygen.restype = None
        L.ref_keygen.argtypes = [u64, u64, u64p, C.POINTER(i64), C.POINTER(i64)]
        L.ref_encrypt_seq.restype = None
        L.ref_encrypt_seq.argtypes = [u64, u64, u64p, C.c_uint, u8p, u64, u64p, u64p]
        for name in ("ref_mul", "ref_mul_inplace", "ref_add", "ref_add_inplace"):
            f = getattr(L, name)
            f.restype = u64
            f.argtypes = [u64, u64, u64p, u64p, u64, u64p, u64p, u64, u64p, u64p]
        L.ref_decrypt.restype = C.c_uint
        L.ref_decrypt.argtypes = [u64, u64, u64p, u64p, u64p, u64]
        L.ref_perm_random.restype = None
        L.ref_perm_random.argtypes = [u64, C.c_uint, u64p]
        L.ref_perm_inverse.restype = None
        L.ref_perm_inverse.argtypes = [u64p, u64, u64p]
        L.ref_perm_compose.restype = u64
        L.ref_perm_compose.argtypes = [u64p, u64, u64p, u64, u64p]
        L.ref_permute_ciphertext.restype = u64
        L.ref_permute_ciphertext.argtypes = [u64, u64, u64p, u64p, u64p, u64, u64p, u64p]
        L.ref_permute_key.restype = None
        L.ref_permute_key.argtypes = [u64, u64, u64p, u64p, u64p]
        L.ref_time_mul.restype = C.c_double
        L.ref_time_mul.argtypes = [u64, u64, u64p, u64, u64p, u64, u64, u64p]
        L.ref_time_decrypt.restype = C.c_double
        L.ref_time_decrypt.argtypes = [u64, u64, u64p, u64p, u64, u64, u64p]
        L.ref_time_circuit.restype = C.c_double
        L.ref_time_circuit.argtypes = [u64, u64, C.c_uint, u64, C.c_uint, C.POINTER(u64), C.POINTER(C.c_uint)]
        L.ref_text.restype = u64
        L.ref_text.argtypes = [C.c_int, u64, u64, u64p, u64p, u64, C.c_char_p, u64]

    def default_len(self, n: int, d: int) -> int:
        return int(self.lib.ref_default_len(n, d))

    def context_s(self, n: int, d: int) -> int:
        return int(self.lib.ref_context_s(n, d))

    def keygen(self, n: int, d: int):
        key = np.zeros(d, dtype=np.uint64)
        t0, t1 = i64(0), i64(0)
        self.lib.ref_keygen(n, d, _p64(key), C.byref(t0), C.byref(t1))
        return key, int(t0.value), int(t1.value)

    def encrypt_seq(self, n: int, d: int, key, seed: int, bits):
        key = as_u64(key)
        bits = np.ascontiguousarray(np.asarray(bits, dtype=np.uint8))
        dl = (n + 63) // 64
        # one spare word: for N%64==0 the reference's own bitlen has no spare either,
        # but our receiving buffers never overflow (copy_out copies getLen() words)
        out = np.zeros(bits.size * dl, dtype=np.uint64)
        bl = np.zeros(bits.size * dl, dtype=np.uint64)
        self.lib.ref_encrypt_seq(n, d, _p64(key), seed & 0xFFFFFFFF, bits.ctypes.data_as(u8p),
                                 bits.size, _p64(out), _p64(bl))
        return out, bl

    def _binop(self, name: str, n: int, d: int, v1, bl1, v2, bl2, outlen: int):
        v1, bl1, v2, bl2 = as_u64(v1), as_u64(bl1), as_u64(v2), as_u64(bl2)
        out = np.zeros(max(outlen, 1), dtype=np.uint64)
        blo = np.zeros(max(outlen, 1), dtype=np.uint64)
        got = int(getattr(self.lib, name)(n, d, _p64(v1), _p64(bl1), v1.size, _p64(v2), _p64(bl2),
                                          v2.size, _p64(out), _p64(blo)))
        assert got == outlen, (name, got, outlen)
        return out[:got], blo[:got]

    def mul(self, n, d, v1, bl1, v2, bl2, inplace=False):
        dl = (n + 63) // 64
        l1, l2 = len(v1), len(v2)
        outlen = l1 if (l1 == dl and l1 == l2) else ((l1 // dl) * l2) // dl * dl
        return self._binop("ref_mul_inplace" if inplace else "ref_mul", n, d, v1, bl1, v2, bl2, outlen)

    def add(self, n, d, v1, bl1, v2, bl2, inplace=False):
        return self._binop("ref_add_inplace" if inplace else "ref_add", n, d, v1, bl1, v2, bl2,
                           len(v1) + len(v2))

    def decrypt(self, n, d, key, v, bl) -> int:
        key, v, bl = as_u64(key), as_u64(v), as_u64(bl)
        return int(self.lib.ref_decrypt(n, d, _p64(key), _p64(v), _p64(bl), v.size))

    def perm_random(self, size: int, seed: int) -> np.ndarray:
        out = np.zeros(size, dtype=np.uint64)
        self.lib.ref_perm_random(size, seed & 0xFFFFFFFF, _p64(out))
        return out

    def perm_inverse(self, perm) -> np.ndarray:
        perm = as_u64(perm)
        out = np.zeros(perm.size, dtype=np.uint64)
        self.lib.ref_perm_inverse(_p64(perm), perm.size, _p64(out))
        return out

    def perm_compose(self, a, b):
        a, b = as_u64(a), as_u64(b)
        out = np.zeros(max(a.size, 1), dtype=np.uint64)
        n = int(self.lib.ref_perm_compose(_p64(a), a.size, _p64(b), b.size, _p64(out)))
        return out[:n] if n else None

    def permute_ciphertext(self, n, d, perm, v, bl):
        perm, v, bl = as_u64(perm), as_u64(v), as_u64(bl)
        dl = (n + 63) // 64
        out = np.zeros(dl, dtype=np.uint64)
        blo = np.zeros(dl, dtype=np.uint64)
        got = int(self.lib.ref_permute_ciphertext(n, d, _p64(perm), _p64(v), _p64(bl), v.size,
                                                  _p64(out), _p64(blo)))
        return out[:got], blo[:got]

    def permute_key(self, n, d, perm, key) -> np.ndarray:
        perm, key = as_u64(perm), as_u64(key)
        out = np.zeros(d, dtype=np.uint64)
        self.lib.ref_permute_key(n, d, _p64(perm), _p64(key), _p64(out))
        return out

    def time_mul(self, n, d, v1, v2, iters: int) -> float:
        v1, v2 = as_u64(v1), as_u64(v2)
        sink = u64(0)
        return float(self.lib.ref_time_mul(n, d, _p64(v1), v1.size, _p64(v2), v2.size, iters,
                                           C.byref(sink)))

    def time_decrypt(self, n, d, key, v, iters: int) -> float:
        key, v = as_u64(key), as_u64(v)
        sink = u64(0)
        return float(self.lib.ref_time_decrypt(n, d, _p64(key), _p64(v), v.size, iters, C.byref(sink)))

    def time_circuit(self, n: int, d: int, levels: int, iters: int, seed: int = 1):
        """(seconds, result terms, decrypted bit) of `iters` depth-`levels` config-5 circuits."""
        terms, bit = u64(0), C.c_uint(0)
        t = float(self.lib.ref_time_circuit(n, d, levels, iters, seed, C.byref(terms), C.byref(bit)))
        return t, int(terms.value), int(bit.value)

    TEXT_KINDS = {"ciphertext": 0, "key": 1, "context": 2, "plaintext": 3, "permutation": 4}

    def text(self, kind: str, n: int, d: int, a=None, b=None, length: int = 0) -> str:
        """operator<< output of one object (see ref_text in ref_driver.cpp)."""
        a = as_u64(a) if a is not None else np.zeros(1, dtype=np.uint64)
        b = as_u64(b) if b is not None else np.zeros(1, dtype=np.uint64)
        k = self.TEXT_KINDS[kind]
        need = int(self.lib.ref_text(k, n, d, _p64(a), _p64(b), length, None, 0))
        buf = C.create_string_buffer(need + 1)
        self.lib.ref_text(k, n, d, _p64(a), _p64(b), length, buf, need)
        return buf.raw[:need].decode("ascii")


# ---- text forms (operator<<): restated here because they are pure formatting -------------
def text_ciphertext(v, bitlen) -> str:
    """src/Ciphertext.cpp:185-202: per word, its top bitlen[word] bits MSB-first; one newline."""
    v, bitlen = as_u64(v), as_u64(bitlen)
    return "".join(format(int(w), "064b")[: int(b)] for w, b in zip(v, bitlen)) + "\n"


def text_key(key) -> str:
    """src/SecretKey.cpp:22-29: indices in stored order, each followed by a blank."""
    return "".join("%d " % int(k) for k in as_u64(key)) + "\n"


def text_context(n: int, d: int) -> str:
    """src/Context.cpp:40-47 (S = N/(2D), src/Context.cpp:20-29)."""
    return "N= %d\nD= %d\nS= %d\n" % (n, d, n // (2 * d))


def text_plaintext(bit: int) -> str:
    """src/Plaintext.cpp:10-19."""
    return "%d\n" % (bit & 1)


def text_permutation(perm) -> str:
    """src/Permutation.cpp:33-46: two-row notation."""
    perm = as_u64(perm)
    top = "".join("%d " % i for i in range(perm.size))
    bot = "".join("%d " % int(x) for x in perm)
    return "(%s)\n(%s)\n" % (top, bot)


def load_ref() -> Optional[Ref]:
    return Ref(REF_SO) if os.path.exists(REF_SO) else None
