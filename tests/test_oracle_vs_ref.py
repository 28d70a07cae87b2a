"""Pin the CPU restatement (oracle/csgn_oracle.c) against the GENUINE reference
(oracle/_ref/libcsgn_ref.so = /root/reference/src compiled by oracle/Makefile).

Mirrors the reference's own demo programs (tests/basic_operations.cpp,
tests/permutations.cpp) but with assertions, plus the edge cases listed in SURVEY 8c.
Skips where oracle/_ref is not available; the committed golden vectors
(test_oracle_golden.py) carry the same pins everywhere else.
"""
import time

import numpy as np
import pytest

from oracle.binding import (canonical_bitlen, glibc_draws, text_ciphertext, text_context, text_key,
                            text_permutation, text_plaintext)

CONTEXTS = [(1247, 16), (4096, 32), (63, 4), (64, 4), (65, 4), (128, 8), (130, 5), (100, 1)]


def make_key(n, d, seed):
    rng = np.random.default_rng(seed)
    return rng.permutation(n)[:d].astype(np.uint64)


def rand_terms(oracle, n, terms, seed):
    return oracle.synth(seed, n, 0, terms * oracle.default_len(n))


@pytest.mark.parametrize("n,d", CONTEXTS)
def test_context(oracle, ref, n, d):
    assert oracle.default_len(n) == ref.default_len(n, d)
    assert oracle.context_s(n, d) == ref.context_s(n, d)


@pytest.mark.parametrize("n,d", CONTEXTS)
@pytest.mark.parametrize("seed", [1, 7, 12345])
def test_encrypt_stream_matches_reference(oracle, ref, n, d, seed):
    """Bit-exact fresh ciphertexts from one srand(seed) stream, bits 1,0,0,1,0,1,1,0."""
    key = make_key(n, d, seed)
    bits = [1, 0, 0, 1, 0, 1, 1, 0]
    want, want_bl = ref.encrypt_seq(n, d, key, seed, bits)
    draws = glibc_draws(seed, (n + 2) * len(bits))
    got, used = oracle.encrypt_seq(n, key, bits, draws)
    assert np.array_equal(got, want)
    assert np.array_equal(oracle.bitlen(n, len(bits))[: want_bl.size], want_bl) or n % 64 == 0
    # padding bits of the last word stay zero
    rem = n % 64
    if rem:
        dl = oracle.default_len(n)
        assert not np.any(got.reshape(-1, dl)[:, -1] & np.uint64((1 << (64 - rem)) - 1))
    # decrypt agrees with the plaintext for every fresh ciphertext (d==1 excepted: the
    # reference's Enc(0) is random there, see oracle/csgn_oracle.c)
    dl = oracle.default_len(n)
    bl = canonical_bitlen(n, 1)
    for i, b in enumerate(bits):
        ct = got[i * dl:(i + 1) * dl]
        r = ref.decrypt(n, d, key, ct, bl)
        assert oracle.decrypt(n, key, ct) == r
        assert oracle.decrypt_canonical(n, key, ct) == r
        if d > 1:
            assert r == b


@pytest.mark.parametrize("n,d", [(1247, 16), (4096, 32), (65, 4), (64, 4)])
@pytest.mark.parametrize("t1,t2", [(1, 1), (1, 2), (2, 1), (2, 2), (3, 5), (7, 1), (1, 7), (32, 32)])
def test_mul_matches_reference(oracle, ref, n, d, t1, t2):
    a = rand_terms(oracle, n, t1, 11)
    b = rand_terms(oracle, n, t2, 22)
    # distinct bitlen patterns on each side expose the left-operand rule (Ciphertext.cpp:172)
    bl1 = (np.arange(a.size, dtype=np.uint64) % 60) + 1
    bl2 = (np.arange(b.size, dtype=np.uint64) % 50) + 5
    want, want_bl = ref.mul(n, d, a, bl1, b, bl2)
    got, got_bl = oracle.mul(n, a, b, bl1)
    assert np.array_equal(got, want)
    assert np.array_equal(got_bl, want_bl)
    want2, want_bl2 = ref.mul(n, d, a, bl1, b, bl2, inplace=True)
    assert np.array_equal(got, want2) and np.array_equal(got_bl, want_bl2)


@pytest.mark.parametrize("n,d", [(1247, 16), (4096, 32), (65, 4)])
@pytest.mark.parametrize("t1,t2", [(1, 1), (1, 3), (4, 2), (17, 9)])
def test_add_matches_reference(oracle, ref, n, d, t1, t2):
    a = rand_terms(oracle, n, t1, 5)
    b = rand_terms(oracle, n, t2, 6)
    bl1, bl2 = canonical_bitlen(n, t1), canonical_bitlen(n, t2)
    want, want_bl = ref.add(n, d, a, bl1, b, bl2)
    got, got_bl = oracle.add(a, b, bl1, bl2)
    assert np.array_equal(got, want) and np.array_equal(got_bl, want_bl)
    want2, want_bl2 = ref.add(n, d, a, bl1, b, bl2, inplace=True)
    assert np.array_equal(got, want2) and np.array_equal(got_bl, want_bl2)


@pytest.mark.parametrize("n,d", [(1247, 16), (4096, 32), (65, 4), (64, 4), (63, 4)])
def test_decrypt_multiterm_matches_reference(oracle, ref, n, d):
    """XOR over terms of AND over key, on term lists with a controlled number of hits."""
    key = make_key(n, d, 3)
    mask = oracle.key_mask(n, key)
    dl = oracle.default_len(n)
    for terms, hits in [(1, 0), (1, 1), (2, 1), (2, 2), (5, 3), (64, 17), (257, 100)]:
        v = rand_terms(oracle, n, terms, 100 + terms).reshape(terms, dl)
        v[:hits] |= mask          # force `hits` terms to satisfy the key
        # make sure the others do not hit by clearing one secret position
        w, b = int(key[0]) // 64, 63 - int(key[0]) % 64
        v[hits:, w] &= ~np.uint64(1 << b)
        flat = np.ascontiguousarray(v.reshape(-1))
        bl = canonical_bitlen(n, terms)
        want = ref.decrypt(n, d, key, flat, bl)
        assert want == hits % 2
        assert oracle.decrypt(n, key, flat, bl) == want
        assert oracle.decrypt(n, key, flat) == want
        assert oracle.decrypt_canonical(n, key, flat) == want


def test_basic_operations_program(oracle, ref):
    """tests/basic_operations.cpp with assertions: Dec(Enc(1)+Enc(0))=1, Dec(Enc(1)*Enc(0))=0."""
    n, d = 1247, 16
    for seed in range(20):
        key = make_key(n, d, seed)
        cts, _ = ref.encrypt_seq(n, d, key, seed, [1, 0])
        dl = oracle.default_len(n)
        c1, c0 = cts[:dl], cts[dl:]
        bl = canonical_bitlen(n, 1)
        added, added_bl = oracle.add(c1, c0, bl, bl)
        mult, mult_bl = oracle.mul(n, c1, c0, bl)
        r_add, r_add_bl = ref.add(n, d, c1, bl, c0, bl)
        r_mul, r_mul_bl = ref.mul(n, d, c1, bl, c0, bl)
        assert np.array_equal(added, r_add) and np.array_equal(mult, r_mul)
        assert ref.decrypt(n, d, key, r_add, r_add_bl) == 1 == oracle.decrypt(n, key, added, added_bl)
        assert ref.decrypt(n, d, key, r_mul, r_mul_bl) == 0 == oracle.decrypt(n, key, mult, mult_bl)


@pytest.mark.parametrize("size", [1, 2, 17, 64, 65, 1247])
@pytest.mark.parametrize("seed", [1, 99])
def test_permutation_generation_inverse_compose(oracle, ref, size, seed):
    want = ref.perm_random(size, seed)
    draws = glibc_draws(seed, 64 * size + 1000 if size < 200 else 40 * size)
    got, _ = oracle.perm_random(size, draws)
    assert np.array_equal(got, want)
    assert sorted(got.tolist()) == list(range(size))
    inv = oracle.perm_inverse(got)
    assert np.array_equal(inv, ref.perm_inverse(got))
    comp = oracle.perm_compose(got, inv)
    assert np.array_equal(comp, ref.perm_compose(got, inv))
    assert np.array_equal(comp, np.arange(size, dtype=np.uint64))
    assert oracle.perm_compose(got, inv[:-1]) is None and ref.perm_compose(got, inv[:-1]) is None


@pytest.mark.parametrize("n,d", [(1247, 16), (65, 4), (63, 4), (130, 5)])
def test_permute_ciphertext_and_key(oracle, ref, n, d):
    """tests/permutations.cpp with assertions, plus the multi-term truncation."""
    seed = 5
    key = make_key(n, d, seed)
    perm = ref.perm_random(n, seed)
    cts, _ = ref.encrypt_seq(n, d, key, seed, [1, 0, 1])
    dl = oracle.default_len(n)
    bl1 = canonical_bitlen(n, 1)
    pkey = oracle.permute_key(n, perm, key)
    assert np.array_equal(pkey, ref.permute_key(n, d, perm, key))
    for i, bit in enumerate([1, 0, 1]):
        ct = cts[i * dl:(i + 1) * dl]
        want, want_bl = ref.permute_ciphertext(n, d, perm, ct, bl1)
        got = oracle.permute_ciphertext(n, perm, ct, bl1)
        assert np.array_equal(got, want)
        assert np.array_equal(want_bl, bl1)
        assert ref.decrypt(n, d, pkey, want, want_bl) == bit
        assert oracle.decrypt(n, pkey, got) == bit
    # multi-term input: the reference returns ONE term = permuted first term
    want, _ = ref.permute_ciphertext(n, d, perm, cts, canonical_bitlen(n, 3))
    got = oracle.permute_ciphertext(n, perm, cts)
    assert want.size == dl and np.array_equal(got, want)
    assert np.array_equal(got, oracle.permute_ciphertext(n, perm, cts[:dl]))


def _explained_by_stale_tail(draws, n, key):
    """The reference tests each draw against the whole, partly UNINITIALISED key array
    (SecretKey.cpp:318-327, Helpers.cpp:18-24), so a stale heap word can reject a draw the
    restatement accepts.  ref_keygen conditions the heap so this should not happen; if it still
    does, accept a key that is the draw sequence with only such rejections: every skipped draw
    is either a repeat of an accepted index or differs from the index accepted next, and there
    are at most d unexplained skips."""
    key = key.tolist()
    count, unexplained = 0, 0
    for v in (int(x) % n for x in draws):
        if count == len(key):
            break
        if v in key[:count]:
            continue
        if v == key[count]:
            count += 1
        else:
            unexplained += 1
    return count == len(key) and unexplained <= len(key)


def test_keygen_restatement_reproduces_reference_key(oracle, ref):
    """The reference seeds keygen from time(NULL) (SecretKey.cpp:311-312); recover the seed
    from the [t_before, t_after] window and check the restated sampler yields the same key."""
    for n, d in [(1247, 16), (4096, 32), (65, 4)]:
        key, t0, t1 = ref.keygen(n, d)
        assert len(set(key.tolist())) == d and int(key.max()) < n
        hit = False
        for t in range(t0 - 1, t1 + 2):
            draws = glibc_draws(t, 64 * d + 64)
            cand, _ = oracle.keygen(n, d, draws)
            if np.array_equal(cand, key) or _explained_by_stale_tail(draws, n, key):
                hit = True
                break
        assert hit, "restated keygen did not reproduce the reference key for any seed in window"


def test_large_products_digest(oracle, ref):
    """32x32 and 256x256 products: full compare (the 1024x1024 digest lives in golden/)."""
    n, d = 1247, 16
    for t in (32, 256):
        a = rand_terms(oracle, n, t, 1000 + t)
        b = rand_terms(oracle, n, t, 2000 + t)
        bl = canonical_bitlen(n, t)
        want, _ = ref.mul(n, d, a, bl, b, bl)
        got, _ = oracle.mul(n, a, b)
        assert np.array_equal(got, want)
        assert oracle.digest(got) == oracle.digest(want)


@pytest.mark.parametrize("n,d", [(1247, 16), (4096, 32), (65, 4), (63, 4)])
def test_text_forms_match_reference(oracle, ref, n, d):
    """operator<< of every class (text dump parity)."""
    key = make_key(n, d, 9)
    cts, bl = ref.encrypt_seq(n, d, key, 9, [1, 0, 1])
    dl = oracle.default_len(n)
    want = ref.text("ciphertext", n, d, cts, canonical_bitlen(n, 3), cts.size)
    assert want == text_ciphertext(cts, canonical_bitlen(n, 3))
    assert len(want) == 3 * n + 1
    # a product keeps the left operand's Bitlen; the dump follows Bitlen word by word
    prod, pbl = ref.mul(n, d, cts[:2 * dl], canonical_bitlen(n, 2), cts[dl:], canonical_bitlen(n, 2))
    assert ref.text("ciphertext", n, d, prod, pbl, prod.size) == text_ciphertext(prod, pbl)
    odd_bl = (np.arange(dl, dtype=np.uint64) % 64) + 1
    assert ref.text("ciphertext", n, d, cts[:dl], odd_bl, dl) == text_ciphertext(cts[:dl], odd_bl)
    assert ref.text("key", n, d, key, None, d) == text_key(key)
    assert ref.text("context", n, d) == text_context(n, d)
    assert ref.text("plaintext", n, d, length=1) == text_plaintext(1) == "1\n"
    assert ref.text("plaintext", n, d, length=0) == text_plaintext(0) == "0\n"
    perm = ref.perm_random(n, 4)
    assert ref.text("permutation", n, d, perm, None, n) == text_permutation(perm)


@pytest.mark.parametrize("n,d", [(1247, 16), (4096, 32), (65, 4), (130, 5), (200, 6)])
def test_noncanonical_bitlen_stream_semantics(oracle, ref, n, d):
    """decrypt / permutation of ciphertexts whose Bitlen is not the canonical pattern: restatement
    vs the genuine reference on random patterns that keep every addressed position inside the
    stream (src/SecretKey.cpp:104-147, src/Ciphertext.cpp:16-69)."""
    rng = np.random.default_rng(n * 7 + d)
    dl = oracle.default_len(n)
    for terms in (1, 2, 6, 17):
        key = rng.permutation(n)[:d].astype(np.uint64)
        v = oracle.synth(int(rng.integers(1, 1 << 30)), n, 0, terms * dl)
        bl = np.full(terms * dl, 64, dtype=np.uint64)
        slack = int(64 * terms * dl - n * terms)
        cut = rng.integers(0, 4, size=terms * dl)
        while int(cut.sum()) > slack:
            cut[rng.integers(0, cut.size)] = 0
        bl -= cut.astype(np.uint64)
        assert ref.decrypt(n, d, key, v, bl) == oracle.decrypt(n, key, v, bl)
        perm = ref.perm_random(n, int(rng.integers(1, 1 << 20)))
        want, wbl = ref.permute_ciphertext(n, d, perm, v, bl)
        assert np.array_equal(oracle.permute_ciphertext(n, perm, v, bl), want)
