"""Torch-backed plumbing over the C ABI: device buffers, streams, batch calls.

PyTorch is used here ONLY for HBM allocation (`torch.empty(..., device="cuda")`), the current
HIP stream and `torch.distributed`; every arithmetic result comes from libcsgn_hip.so through
`csgn_amd.capi`.  Term words are uint64 on the wire; torch holds them as int64 (same bits).

All methods enqueue on torch's current stream, so `torch.cuda.Event` brackets them correctly.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import capi
from .capi import check


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


class HipPath:
    """One GPU's view of the hot path.  Construct one per process / device."""

    def __init__(self, device: int = 0):
        self.lib = capi.load_library()
        if not torch.cuda.is_available():
            raise capi.CsgnError(capi.CSGN_ERR_NO_DEVICE,
                                 "torch sees no GPU; csgn_amd has no CPU fallback")
        torch.cuda.set_device(device)
        check(self.lib.csgn_init(device))
        self.device = torch.device("cuda", device)

    def close(self) -> None:
        """Destroys the calling thread's csgn_mul_plan handle (its device block goes back to the driver).  Handles of
        other threads are destroyed when those threads call close(), or with the process."""
        plans = getattr(self, "_plans", None)
        handle = getattr(plans, "handle", None) if plans is not None else None
        if handle is not None:
            self.lib.csgn_mul_plan_destroy(handle)
            plans.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing -------------------------------------------------------------------
    @property
    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def empty_words(self, n_words: int) -> torch.Tensor:
        return torch.empty(int(n_words), dtype=torch.int64, device=self.device)

    def upload(self, a: np.ndarray) -> torch.Tensor:
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint64:
            return torch.from_numpy(a.view(np.int64)).to(self.device)
        if a.dtype == np.uint32:
            return torch.from_numpy(a.view(np.int32)).to(self.device)
        return torch.from_numpy(a).to(self.device)

    @staticmethod
    def download(t: torch.Tensor) -> np.ndarray:
        a = t.detach().cpu().numpy()
        if a.dtype == np.int64:
            return a.view(np.uint64)
        if a.dtype == np.int32:
            return a.view(np.uint32)
        return a

    def default_len(self, n_bits: int) -> int:
        return int(self.lib.csgn_default_len(n_bits))

    def key_mask(self, n_bits: int, key) -> np.ndarray:
        key = np.ascontiguousarray(np.asarray(key, dtype=np.uint64))
        mask = np.zeros(self.default_len(n_bits), dtype=np.uint64)
        check(self.lib.csgn_key_mask(n_bits, key.ctypes.data, key.size, mask.ctypes.data))
        return mask

    # -- multiply -------------------------------------------------------------------
    def mul_uniform(self, n_bits: int, batch: int, t1: int, t2: int, left: torch.Tensor,
                    right: torch.Tensor, out: Optional[torch.Tensor] = None,
                    out_slots: int = 0) -> torch.Tensor:
        dl = self.default_len(n_bits)
        assert left.numel() >= batch * t1 * dl and right.numel() >= batch * t2 * dl
        slots = batch if out_slots == 0 else min(out_slots, batch)
        if out is None:
            out = self.empty_words(slots * t1 * t2 * dl)
        assert out.numel() >= slots * t1 * t2 * dl
        check(self.lib.csgn_mul_uniform(n_bits, batch, t1, t2, _ptr(left), _ptr(right), _ptr(out),
                                        out_slots, self.stream))
        return out

    def mul_ragged(self, n_bits: int, left: torch.Tensor, off_left: torch.Tensor,
                   right: torch.Tensor, off_right: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Plan (csgn_mul_plan_ragged, synchronises) + multiply by that plan (csgn_mul_planned)."""
        batch = off_left.numel() - 1
        dl = self.default_len(n_bits)
        off_out = self.empty_words(batch + 1)
        plan = (C.c_uint64 * 4)()
        # one plan object per path AND host thread (a csgn_mul_plan is used by one thread at a time); its
        # device block is grow-only, so the plan costs no allocation after the first call
        import threading
        if getattr(self, "_plans", None) is None:
            self._plans = threading.local()
        if getattr(self._plans, "handle", None) is None:
            self._plans.handle = self.mul_plan()
        handle = self._plans.handle
        check(self.lib.csgn_mul_plan_ragged(handle, batch, _ptr(off_left), _ptr(off_right), _ptr(off_out),
                                            C.byref(plan), self.stream))
        total = int(plan[0])
        out = self.empty_words(max(total * dl, 1))
        check(self.lib.csgn_mul_planned(handle, n_bits, _ptr(left), _ptr(right), _ptr(out), self.stream))
        return out[: total * dl], off_out

    def mul_plan(self) -> C.c_void_p:
        """A csgn_mul_plan object (destroy with lib.csgn_mul_plan_destroy)."""
        handle = C.c_void_p()
        check(self.lib.csgn_mul_plan_create(C.byref(handle)))
        return handle

    def mul_ragged_async(self, n_bits: int, left: torch.Tensor, off_left: torch.Tensor, right: torch.Tensor,
                         off_right: torch.Tensor, capacity_terms: int, out: Optional[torch.Tensor] = None,
                         off_out: Optional[torch.Tensor] = None, plan: Optional[torch.Tensor] = None):
        """csgn_mul_ragged_async: nothing is read back.  Returns (out, off_out, d_plan); mul_ragged_async_result(d_plan)
        fetches the plan numbers and the does-not-fit flag later."""
        batch = off_left.numel() - 1
        dl = self.default_len(n_bits)
        if off_out is None:
            off_out = self.empty_words(batch + 1)
        if out is None:
            out = self.empty_words(max(capacity_terms * dl, 1))
        if plan is None:
            plan = self.empty_words(int(self.lib.csgn_mul_ragged_async_plan_words(batch)))
        check(self.lib.csgn_mul_ragged_async(n_bits, batch, _ptr(left), _ptr(off_left), _ptr(right), _ptr(off_right),
                                             _ptr(out), _ptr(off_out), capacity_terms, _ptr(plan), self.stream))
        return out, off_out, plan

    def mul_ragged_async_result(self, plan: torch.Tensor):
        res = (C.c_uint64 * 5)()
        check(self.lib.csgn_mul_ragged_async_result(_ptr(plan), C.byref(res), self.stream))
        return [int(x) for x in res]

    # -- add ------------------------------------------------------------------------
    def add_uniform(self, n_bits: int, batch: int, t1: int, t2: int, left: torch.Tensor,
                    right: torch.Tensor) -> torch.Tensor:
        dl = self.default_len(n_bits)
        out = self.empty_words(max(batch * (t1 + t2) * dl, 1))
        check(self.lib.csgn_add_uniform(n_bits, batch, t1, t2, _ptr(left), _ptr(right), _ptr(out),
                                        self.stream))
        return out[: batch * (t1 + t2) * dl]

    def add_ragged(self, n_bits: int, left: torch.Tensor, off_left: torch.Tensor,
                   right: torch.Tensor, off_right: torch.Tensor,
                   total_terms_out: Optional[int] = None, max_t1: int = 0, max_t2: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """max_t1 / max_t2: upper bounds on one element's terms if the caller knows them (csgn_add_ragged_bounded)."""
        batch = off_left.numel() - 1
        dl = self.default_len(n_bits)
        if total_terms_out is None:
            total_terms_out = int(self.download(off_left[-1:])[0]) + int(self.download(off_right[-1:])[0])
        total = total_terms_out
        out = self.empty_words(max(total * dl, 1))
        off_out = self.empty_words(batch + 1)
        check(self.lib.csgn_add_ragged_bounded(n_bits, batch, max_t1, max_t2, _ptr(left), _ptr(off_left), _ptr(right),
                                               _ptr(off_right), _ptr(out), _ptr(off_out), total, self.stream))
        return out[: total * dl], off_out

    # -- decrypt ----------------------------------------------------------------------
    def decrypt_uniform(self, n_bits: int, batch: int, terms: int, words: torch.Tensor,
                        mask: torch.Tensor) -> torch.Tensor:
        bits = torch.empty(max(batch, 1), dtype=torch.uint8, device=self.device)
        scratch = torch.empty(int(self.lib.csgn_decrypt_scratch_bytes(batch, batch * terms)),
                              dtype=torch.uint8, device=self.device)
        check(self.lib.csgn_decrypt_uniform(n_bits, batch, terms, _ptr(words), _ptr(mask), _ptr(bits),
                                            _ptr(scratch), self.stream))
        return bits[:batch]

    def decrypt_ragged(self, n_bits: int, words: torch.Tensor, off: torch.Tensor,
                       mask: torch.Tensor, total_terms: Optional[int] = None, max_terms: int = 0) -> torch.Tensor:
        """max_terms: an upper bound on the terms of one ciphertext if the caller knows one (csgn_decrypt_ragged_bounded)."""
        batch = off.numel() - 1
        if total_terms is None:
            total_terms = int(self.download(off[-1:])[0])
        bits = torch.empty(max(batch, 1), dtype=torch.uint8, device=self.device)
        scratch = torch.empty(int(self.lib.csgn_decrypt_scratch_bytes(batch, total_terms)),
                              dtype=torch.uint8, device=self.device)
        check(self.lib.csgn_decrypt_ragged_bounded(n_bits, batch, total_terms, max_terms, _ptr(words), _ptr(off),
                                                   _ptr(mask), _ptr(bits), _ptr(scratch), self.stream))
        return bits[:batch]

    def decrypt_combined_uniform(self, n_bits: int, batch: int, t1: int, t2: int, left: torch.Tensor,
                                 right: torch.Tensor, mask: torch.Tensor, product: bool) -> torch.Tensor:
        """Dec(L*R) (product=True) or Dec(L+R) without materialising the result."""
        bits = torch.empty(max(batch, 1), dtype=torch.uint8, device=self.device)
        scratch = torch.empty(int(self.lib.csgn_decrypt_combined_scratch_bytes(batch, t1, t2)),
                              dtype=torch.uint8, device=self.device)
        fn = self.lib.csgn_decrypt_product_uniform if product else self.lib.csgn_decrypt_sum_uniform
        check(fn(n_bits, batch, t1, t2, _ptr(left), _ptr(right), _ptr(mask), _ptr(bits), _ptr(scratch),
                 self.stream))
        return bits[:batch]

    def compact_ragged(self, n_bits: int, words: torch.Tensor, off: torch.Tensor,
                       total_terms: Optional[int] = None, max_terms: int = 0,
                       out: Optional[torch.Tensor] = None, off_out: Optional[torch.Tensor] = None,
                       scratch: Optional[torch.Tensor] = None,
                       sync: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """Extension: mod-2 compaction (identical terms cancel in pairs).  Returns (terms, CSR offsets).
        max_terms: upper bound on any one ciphertext's terms if known (0 = unknown).  With sync=False
        nothing is read back and `out` is returned whole (benchmarks)."""
        batch = off.numel() - 1
        dl = self.default_len(n_bits)
        if total_terms is None:
            total_terms = int(self.download(off[-1:])[0])
        if out is None:
            out = self.empty_words(max(total_terms * dl, 1))
        if off_out is None:
            off_out = self.empty_words(batch + 1)
        if scratch is None:
            scratch = torch.empty(int(self.lib.csgn_compact_scratch_bytes(n_bits, batch, total_terms)),
                                  dtype=torch.uint8, device=self.device)
        check(self.lib.csgn_compact_ragged(n_bits, batch, total_terms, max_terms, _ptr(words), _ptr(off),
                                           _ptr(out), _ptr(off_out), _ptr(scratch), self.stream))
        if not sync:
            return out, off_out
        kept = int(self.download(off_out[-1:])[0])
        return out[: kept * dl], off_out

    # -- encrypt ----------------------------------------------------------------------
    def encrypt_explicit(self, n_bits: int, d: int, plain: torch.Tensor, rnd: torch.Tensor,
                         chosen: torch.Tensor, last: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        batch = plain.numel()
        out = self.empty_words(max(batch * self.default_len(n_bits), 1))
        check(self.lib.csgn_encrypt_explicit(n_bits, d, batch, _ptr(plain), _ptr(rnd), _ptr(chosen),
                                             _ptr(last), _ptr(mask), _ptr(out), self.stream))
        return out[: batch * self.default_len(n_bits)]

    def encrypt_device_rng(self, n_bits: int, d: int, plain: torch.Tensor, key: torch.Tensor,
                           mask: torch.Tensor, seed: int) -> torch.Tensor:
        batch = plain.numel()
        out = self.empty_words(max(batch * self.default_len(n_bits), 1))
        check(self.lib.csgn_encrypt_device_rng(n_bits, d, batch, _ptr(plain), _ptr(key), _ptr(mask),
                                               seed & (2**64 - 1), _ptr(out), self.stream))
        return out[: batch * self.default_len(n_bits)]

    def rng_from_seed(self, seed: int, rounds: int = 8) -> "capi.CsgnRng":
        r = capi.CsgnRng()
        check(self.lib.csgn_rng_from_seed(C.byref(r), seed & (2**64 - 1), rounds))
        return r

    def rng_from_os(self, rounds: int = 8) -> "capi.CsgnRng":
        r = capi.CsgnRng()
        check(self.lib.csgn_rng_from_os(C.byref(r), rounds))
        return r

    def encrypt_keyed(self, n_bits: int, d: int, plain: torch.Tensor, key: torch.Tensor, mask: torch.Tensor,
                      rng: "capi.CsgnRng", first_ciphertext: int = 0,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Keyed (ChaCha) device-generator encrypt; ciphertext i draws stream position first_ciphertext + i."""
        batch = plain.numel()
        dl = self.default_len(n_bits)
        if out is None:
            out = self.empty_words(max(batch * dl, 1))
        check(self.lib.csgn_encrypt_keyed(n_bits, d, batch, first_ciphertext, _ptr(plain), _ptr(key), _ptr(mask),
                                          C.byref(rng), _ptr(out), self.stream))
        return out[: batch * dl]

    def encrypt_mul_keyed(self, n_bits: int, d: int, plain_a: torch.Tensor, plain_b: torch.Tensor, key: torch.Tensor,
                          mask: torch.Tensor, rng_a: "capi.CsgnRng", rng_b: "capi.CsgnRng", first_ciphertext: int = 0,
                          with_bits: bool = True):
        """Fused fresh chain: product b = Enc_A(plain_a[b]) & Enc_B(plain_b[b]) in one kernel
        (csgn_encrypt_mul_keyed); returns (products, Dec bits or None)."""
        batch = plain_a.numel()
        assert plain_b.numel() == batch
        dl = self.default_len(n_bits)
        out = self.empty_words(max(batch * dl, 1))
        bits = torch.full((max(batch, 1),), 7, dtype=torch.uint8, device=self.device) if with_bits else None
        check(self.lib.csgn_encrypt_mul_keyed(n_bits, d, batch, first_ciphertext, _ptr(plain_a), _ptr(plain_b), _ptr(key),
                                              _ptr(mask), C.byref(rng_a), C.byref(rng_b), _ptr(out), _ptr(bits),
                                              self.stream))
        return out[: batch * dl], (bits[:batch] if with_bits else None)

    # -- permutation --------------------------------------------------------------------
    def permute_uniform(self, n_bits: int, batch: int, terms_in: int, words: torch.Tensor,
                        perm: torch.Tensor, per_term: bool = False) -> torch.Tensor:
        dl = self.default_len(n_bits)
        n_out = batch * (terms_in if per_term else 1) * dl
        out = self.empty_words(max(n_out, 1))
        check(self.lib.csgn_permute_uniform(n_bits, batch, terms_in, int(per_term), _ptr(words),
                                            _ptr(perm), _ptr(out), self.stream))
        return out[:n_out]

    # -- explicit bitlen (one ciphertext) ------------------------------------------------
    def decrypt_bitlen(self, n_bits: int, key: torch.Tensor, words: torch.Tensor, bitlen: torch.Tensor) -> int:
        length = words.numel()
        bit = torch.zeros(1, dtype=torch.uint8, device=self.device)
        scratch = torch.empty(int(self.lib.csgn_bitlen_scratch_bytes(length)), dtype=torch.uint8, device=self.device)
        check(self.lib.csgn_decrypt_bitlen(n_bits, key.numel(), length, _ptr(words), _ptr(bitlen), _ptr(key),
                                           _ptr(bit), _ptr(scratch), self.stream))
        return int(bit.item())

    def permute_bitlen(self, n_bits: int, words: torch.Tensor, bitlen: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
        length = words.numel()
        out = self.empty_words(self.default_len(n_bits))
        scratch = torch.empty(int(self.lib.csgn_bitlen_scratch_bytes(length)), dtype=torch.uint8, device=self.device)
        check(self.lib.csgn_permute_bitlen(n_bits, length, _ptr(words), _ptr(bitlen), _ptr(perm), _ptr(out),
                                           _ptr(scratch), self.stream))
        return out

    # -- harness --------------------------------------------------------------------------
    def synth_fill(self, seed: int, n_bits: int, first_word: int, n_words: int,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = self.empty_words(n_words)
        check(self.lib.csgn_synth_fill(seed & (2**64 - 1), n_bits, first_word, n_words, _ptr(out),
                                       self.stream))
        return out

    def digest(self, words: torch.Tensor, n_words: Optional[int] = None, first_index: int = 0) -> int:
        if n_words is None:
            n_words = words.numel()
        acc = torch.zeros(1, dtype=torch.int64, device=self.device)
        check(self.lib.csgn_digest(_ptr(words), n_words, first_index, _ptr(acc), self.stream))
        return int(acc.item()) & (2**64 - 1)
