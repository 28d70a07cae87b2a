"""Pin the CPU restatement against the committed known-answer vectors.

tests/golden/csgn_kat.json was produced from the genuine reference by
tests/golden/gen_golden.py; this test needs neither /root/reference nor oracle/_ref.
"""
import json
import os

import numpy as np
import pytest

from oracle.binding import (canonical_bitlen, glibc_draws, text_ciphertext, text_context, text_key,
                            text_permutation, text_plaintext)

KAT_PATH = os.path.join(os.path.dirname(__file__), "golden", "csgn_kat.json")


def words(hex_list):
    return np.array([int(h, 16) for h in hex_list], dtype=np.uint64)


@pytest.fixture(scope="module")
def kat():
    with open(KAT_PATH) as f:
        return json.load(f)


def test_golden_encrypt(oracle, kat):
    for case in kat["encrypt"]:
        n, d, seed = case["n"], case["d"], case["seed"]
        key = np.array(case["key"], dtype=np.uint64)
        draws = glibc_draws(seed, (n + 2) * len(case["bits"]))
        got, _ = oracle.encrypt_seq(n, key, case["bits"], draws)
        assert np.array_equal(got, words(case["ct"])), (n, d, seed)
        dl = oracle.default_len(n)
        for i, want in enumerate(case["dec"]):
            assert oracle.decrypt(n, key, got[i * dl:(i + 1) * dl]) == want
            assert oracle.decrypt_canonical(n, key, got[i * dl:(i + 1) * dl]) == want


def test_golden_mul(oracle, kat):
    for case in kat["mul"]:
        out, bl = oracle.mul(case["n"], words(case["a"]), words(case["b"]),
                             np.array(case["bitlen_a"], dtype=np.uint64))
        assert np.array_equal(out, words(case["out"])), (case["n"], case["t1"], case["t2"])
        assert np.array_equal(bl, np.array(case["bitlen_out"], dtype=np.uint64))


def test_golden_add(oracle, kat):
    for case in kat["add"]:
        n, t1, t2 = case["n"], case["t1"], case["t2"]
        out, bl = oracle.add(words(case["a"]), words(case["b"]),
                             canonical_bitlen(n, t1), canonical_bitlen(n, t2))
        assert np.array_equal(out, words(case["out"]))
        assert np.array_equal(bl, np.array(case["bitlen_out"], dtype=np.uint64))


def rebuild_decrypt_input(oracle, case):
    n, terms, hits = case["n"], case["terms"], case["hits"]
    key = np.array(case["key"], dtype=np.uint64)
    dl = oracle.default_len(n)
    v = oracle.synth(case["seed"], n, 0, terms * dl).reshape(terms, dl)
    v[:hits] |= oracle.key_mask(n, key)
    w, b = int(key[0]) // 64, 63 - int(key[0]) % 64
    v[hits:, w] &= ~np.uint64(1 << b)
    return key, np.ascontiguousarray(v.reshape(-1))


def test_golden_decrypt(oracle, kat):
    for case in kat["decrypt"]:
        key, flat = rebuild_decrypt_input(oracle, case)
        assert "%016x" % oracle.digest(flat) == case["digest"]
        if case["v"] is not None:
            assert np.array_equal(flat, words(case["v"]))
        assert oracle.decrypt(case["n"], key, flat) == case["bit"]
        assert oracle.decrypt_canonical(case["n"], key, flat) == case["bit"]


def test_golden_permutation(oracle, kat):
    for case in kat["permutation"]:
        n, seed = case["n"], case["seed"]
        key = np.array(case["key"], dtype=np.uint64)
        perm = np.array(case["perm"], dtype=np.uint64)
        got_perm, _ = oracle.perm_random(n, glibc_draws(seed, 64 * n + 1000))
        assert np.array_equal(got_perm, perm)
        inv = oracle.perm_inverse(perm)
        assert np.array_equal(inv, np.array(case["inverse"], dtype=np.uint64))
        assert case["compose_is_identity"]
        assert np.array_equal(oracle.perm_compose(perm, inv), np.arange(n, dtype=np.uint64))
        pkey = oracle.permute_key(n, perm, key)
        assert np.array_equal(pkey, np.array(case["permuted_key"], dtype=np.uint64))
        cts = words(case["ct"])
        dl = oracle.default_len(n)
        first = oracle.permute_ciphertext(n, perm, cts[:dl])
        assert np.array_equal(first, words(case["permuted_first"]))
        multi = oracle.permute_ciphertext(n, perm, cts)
        assert np.array_equal(multi, words(case["permuted_multi"]))
        assert np.array_equal(multi, first)          # the reference's truncation
        assert oracle.decrypt(n, pkey, first) == case["dec_permuted"] == 1


def test_golden_digests(oracle, kat):
    for case in kat["digest"]:
        n, t1, t2 = case["n"], case["t1"], case["t2"]
        dl = oracle.default_len(n)
        a = oracle.synth(case["seed_a"], n, 0, t1 * dl)
        b = oracle.synth(case["seed_b"], n, 0, t2 * dl)
        out, _ = oracle.mul(n, a, b)
        assert out.size == case["out_len"]
        assert "%016x" % oracle.digest(out) == case["digest"]
        assert np.array_equal(out[:4], words(case["first_words"]))
        assert np.array_equal(out[-4:], words(case["last_words"]))


def test_golden_keygen(oracle, kat):
    for case in kat["keygen"]:
        key, used = oracle.keygen(case["n"], case["d"], glibc_draws(case["srand"], 64 * case["d"] + 64))
        assert np.array_equal(key, np.array(case["key"], dtype=np.uint64))
        assert used == case["draws_used"]


def test_golden_basic_operations(oracle, kat):
    c = kat["basic_operations"]
    n = c["n"]
    key = np.array(c["key"], dtype=np.uint64)
    cts, _ = oracle.encrypt_seq(n, key, [1, 0], glibc_draws(c["seed"], 2 * (n + 2)))
    assert np.array_equal(cts[:20], words(c["c1"])) and np.array_equal(cts[20:], words(c["c0"]))
    added, _ = oracle.add(cts[:20], cts[20:])
    mult, _ = oracle.mul(n, cts[:20], cts[20:])
    assert np.array_equal(added, words(c["added"])) and np.array_equal(mult, words(c["multiplied"]))
    assert oracle.decrypt(n, key, added) == c["dec_added"] == 1
    assert oracle.decrypt(n, key, mult) == c["dec_multiplied"] == 0


def test_golden_text_forms(oracle, kat):
    """operator<< strings of every class, as printed by the genuine reference."""
    c = kat["text"]
    n, d = c["n"], c["d"]
    key = np.array(c["key"], dtype=np.uint64)
    cts, _ = oracle.encrypt_seq(n, key, [1, 0], glibc_draws(c["seed"], 2 * (n + 2)))
    assert np.array_equal(cts, words(c["ct"]))
    assert text_ciphertext(cts, canonical_bitlen(n, 2)) == c["ciphertext"]
    assert text_key(key) == c["key_text"]
    assert text_context(n, d) == c["context"]
    assert [text_plaintext(0), text_plaintext(1)] == c["plaintext"]
    perm, _ = oracle.perm_random(n, glibc_draws(c["seed"], 64 * n + 1000))
    assert perm.tolist() == c["perm"]
    assert text_permutation(perm) == c["permutation"]


def test_oracle_homomorphic_properties(oracle):
    """dec(a*b)=dec(a)&dec(b), dec(a+b)=dec(a)^dec(b) on small circuits."""
    n, d = 1247, 16
    rng = np.random.default_rng(9)
    key = rng.permutation(n)[:d].astype(np.uint64)
    draws = glibc_draws(4, 40 * (n + 2))
    bits = [int(b) for b in rng.integers(0, 2, size=32)]
    cts, _ = oracle.encrypt_seq(n, key, bits, draws)
    dl = oracle.default_len(n)
    ct = [cts[i * dl:(i + 1) * dl] for i in range(len(bits))]
    acc, accb = ct[0], bits[0]
    for i in range(1, 12):
        if i % 2:
            acc, _ = oracle.add(acc, ct[i]); accb ^= bits[i]
        else:
            rhs, _ = oracle.add(ct[i], ct[i + 12]); rb = bits[i] ^ bits[i + 12]
            acc, _ = oracle.mul(n, acc, rhs); accb &= rb
        assert oracle.decrypt(n, key, acc) == accb
        assert oracle.decrypt_canonical(n, key, acc) == accb


# ------------------------------------------------- keyed device generator (shared definitions)

def test_chacha_block_known_answer(oracle):
    """RFC 8439 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00.
    In the 64/64 layout used here that is state words 12..15 = 1, 0x09000000, 0x4a000000, 0."""
    key = np.frombuffer(bytes(range(32)), dtype="<u4")
    got = oracle.chacha_block(key, 0x4A000000, 1 | (0x09000000 << 32), 20)
    want = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
            0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert [int(x) for x in got] == want
    # fewer rounds give other words, and the block depends on every input
    assert [int(x) for x in oracle.chacha_block(key, 0x4A000000, 1 | (0x09000000 << 32), 8)] != want
    assert not np.array_equal(oracle.chacha_block(key, 1, 2, 8), oracle.chacha_block(key, 1, 3, 8))
    assert not np.array_equal(oracle.chacha_block(key, 1, 2, 8), oracle.chacha_block(key, 2, 2, 8))


@pytest.mark.parametrize("n,units,passes,group", [(1247, 10, 5, 128), (4096, 32, 1, 8), (64, 1, 1, 256), (65, 1, 1, 256),
                                                  (129, 2, 1, 128), (300, 3, 3, 256), (1300, 11, 11, 256),
                                                  (16384 * 8, 1024, 4, 1)])
def test_keyed_layout(oracle, n, units, passes, group):
    assert oracle.keyed_layout(n) == (units, passes, group)
    assert passes * 256 == group * units


@pytest.mark.parametrize("n,d", [(1247, 16), (4096, 32), (65, 4), (63, 2), (130, 5), (100, 1), (1300, 3)])
def test_keyed_encrypt_restatement_properties(oracle, n, d):
    rng = np.random.default_rng(n + d)
    key = rng.permutation(n)[:d].astype(np.uint64)
    rk, nonce = oracle.rng_from_seed(1234)
    batch = 700
    plain = rng.integers(0, 2, batch).astype(np.uint8)
    dl = oracle.default_len(n)
    ct = oracle.encrypt_keyed(n, key, plain, rk, nonce, 8)
    rem = n % 64
    if rem:
        assert not np.any(ct.reshape(batch, dl)[:, -1] & np.uint64((1 << (64 - rem)) - 1))
    if d > 1:
        for i in range(batch):
            assert oracle.decrypt(n, key, ct[i * dl:(i + 1) * dl]) == plain[i], i
    # the stream is indexed by the GLOBAL ciphertext number: any window reproduces its slice
    for first, cnt in [(0, 1), (1, 5), (127, 3), (128, 200), (255, 257), (699, 1)]:
        win = oracle.encrypt_keyed(n, key, plain[first:first + cnt], rk, nonce, 8, first_ciphertext=first)
        assert np.array_equal(win, ct[first * dl:(first + cnt) * dl]), (first, cnt)
    # other nonce / rounds / key: other words
    assert not np.array_equal(oracle.encrypt_keyed(n, key, plain, rk, nonce + 1, 8), ct)
    assert not np.array_equal(oracle.encrypt_keyed(n, key, plain, rk, nonce, 12), ct)
    rk2 = rk.copy()
    rk2[3] ^= 1
    assert not np.array_equal(oracle.encrypt_keyed(n, key, plain, rk2, nonce, 8), ct)


def test_keyed_plaintext0_rule_has_the_reference_distribution(oracle):
    """csgn_encrypt_keyed draws EVERY position and, when all D secret positions came out 1, clears
    s[draw % D]; the reference (src/SecretKey.cpp:51-76, restated in oracle.encrypt) draws the chosen
    position first and forces it to 0 when all OTHERS are 1.  Exhaustive enumeration over every
    random outcome at N=7: both give exactly the same distribution over ciphertexts."""
    from fractions import Fraction
    from itertools import product
    n = 7
    for key in ([1, 4, 6], [0, 3], [2, 2, 5], [5]):
        d = len(key)
        kset = sorted(set(key))
        # reference: draws = [sRandom] + one per non-chosen position (+ 1 spare)
        ref = {}
        for s_rand in range(d):
            for bits in product((0, 1), repeat=n - 1):
                for spare in (0, 1):
                    draws = np.array([s_rand] + list(bits) + [spare], dtype=np.int32)
                    ct, used = oracle.encrypt(n, key, 0, draws)
                    w = Fraction(1, d * 2 ** (n - 1)) * (Fraction(1, 2) if used == n + 1 else (1 if spare == 0 else 0))
                    if w:
                        ref[int(ct[0])] = ref.get(int(ct[0]), 0) + w
        # keyed rule: N uniform bits, then the clear
        mine = {}
        for bits in product((0, 1), repeat=n):
            word = 0
            for j, b in enumerate(bits):
                word |= b << (63 - j)
            if all(bits[p] for p in kset) and len(kset) >= 2:
                for idx in range(d):
                    w2 = word & ~(1 << (63 - key[idx]))
                    mine[w2] = mine.get(w2, 0) + Fraction(1, d * 2 ** n)
            else:
                mine[word] = mine.get(word, 0) + Fraction(1, 2 ** n)
        assert sum(ref.values()) == 1 and sum(mine.values()) == 1
        assert ref == mine, key


# ------------------------------------- non-canonical bitlen: (v, bitlen) as a bit stream

KAT_BITLEN_PATH = os.path.join(os.path.dirname(__file__), "golden", "csgn_kat_bitlen.json")


def test_golden_bitlen_stream_decrypt_and_permute(oracle):
    """Ciphertexts with a Bitlen other than 64,...,64,N%64 (4-argument constructor / setBitlen):
    decrypt and permutation of the genuine reference, which reads (v, bitlen) as a bit stream
    (src/SecretKey.cpp:104-147, src/Ciphertext.cpp:16-69)."""
    with open(KAT_BITLEN_PATH) as f:
        cases = json.load(f)["bitlen_stream"]
    assert len(cases) >= 10 and {c["dec"] for c in cases} == {0, 1}
    for c in cases:
        n = c["n"]
        key = np.array(c["key"], dtype=np.uint64)
        v, bl = words(c["v"]), np.array(c["bitlen"], dtype=np.uint64)
        assert oracle.decrypt(n, key, v, bl) == c["dec"], (n, c["pattern"])
        got = oracle.permute_ciphertext(n, np.array(c["perm"], dtype=np.uint64), v, bl)
        assert np.array_equal(got, words(c["permuted"])), (n, c["pattern"])
        assert c["permuted_bitlen"] == [int(x) for x in canonical_bitlen(n, 1)] or n % 64 == 0
