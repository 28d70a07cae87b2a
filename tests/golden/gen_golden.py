#!/usr/bin/env python3
"""Generate the committed known-answer vectors in tests/golden/ from the GENUINE reference.

Run in the dev container (where /root/reference exists):

    make -C oracle ref && python tests/golden/gen_golden.py

Every expected value below is produced by oracle/_ref/libcsgn_ref.so, i.e. by the
reference's own sources compiled unmodified (oracle/Makefile) and driven through its
public certFHE:: API (oracle/ref_driver.cpp).  Inputs come from fixed seeds; operands for
mul/add/decrypt are the harness's synthetic words (oracle.synth).  The files
hold DATA ONLY (inputs + expected outputs as hex words) -- no reference source text.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.binding import Oracle, build_ref, canonical_bitlen, glibc_draws, load_ref  # noqa: E402


def hexs(a):
    return ["%016x" % int(x) for x in np.asarray(a, dtype=np.uint64).ravel()]


def ints(a):
    return [int(x) for x in np.asarray(a).ravel()]


def make_key(n, d, seed):
    rng = np.random.default_rng(seed)
    return rng.permutation(n)[:d].astype(np.uint64)


def bitlen_stream_cases(ref, orc):
    """(x) ciphertexts that carry a NON-canonical bitlen side array (4-argument constructor /
    setBitlen): the reference reads (v, bitlen) as a bit stream (src/SecretKey.cpp:104-147,
    src/Ciphertext.cpp:16-69).  Patterns keep every addressed position inside the stream (the
    reference reads out of bounds otherwise).  Written to csgn_kat_bitlen.json."""
    cases = []
    for (n, d, terms, seed) in [(1247, 16, 1, 1), (1247, 16, 7, 2), (4096, 32, 3, 3), (65, 4, 5, 4), (130, 5, 9, 5),
                                (64, 3, 4, 6), (200, 6, 33, 7)]:
        rng = np.random.default_rng(9000 + seed)
        dl = (n + 63) // 64
        key = make_key(n, d, seed)
        v = orc.synth(500 + seed, n, 0, terms * dl)
        # force a few terms to hit so both plaintext values occur
        mask = orc.key_mask(n, key)
        for pattern in ("all64", "mixed"):
            if pattern == "all64":
                bl = np.full(terms * dl, 64, dtype=np.uint64)
            else:
                # per word 64 or a little less, total >= n*terms + margin so every n*k + s[i] is inside
                bl = np.full(terms * dl, 64, dtype=np.uint64)
                slack = int(64 * terms * dl - n * terms)
                cut = rng.integers(0, 3, size=terms * dl)
                while int(cut.sum()) > slack:
                    cut[rng.integers(0, cut.size)] = 0
                bl -= cut.astype(np.uint64)
            vv = v.copy()
            # plant hits: set the stream bits n*k + s[i] for some k (done on the unpacked stream)
            pos = np.concatenate([[0], np.cumsum(bl)]).astype(np.int64)
            for k in range(0, terms, 2):
                for s_i in key:
                    q = n * k + int(s_i)
                    w = int(np.searchsorted(pos, q, side="right") - 1)
                    vv[w] |= np.uint64(1) << np.uint64(63 - (q - pos[w]))
            dec = ref.decrypt(n, d, key, vv, bl)
            perm = ref.perm_random(n, 40 + seed)
            pv, pbl = ref.permute_ciphertext(n, d, perm, vv, bl)
            cases.append(dict(n=n, d=d, terms=terms, pattern=pattern, key=ints(key), v=hexs(vv), bitlen=ints(bl),
                              dec=int(dec), perm=ints(perm), permuted=hexs(pv), permuted_bitlen=ints(pbl)))
    return cases


def main():
    build_ref()
    ref = load_ref()
    if ref is None:
        raise SystemExit("reference not buildable here")
    orc = Oracle()
    if "--bitlen-only" in sys.argv:
        path = os.path.join(HERE, "csgn_kat_bitlen.json")
        with open(path, "w") as f:
            json.dump({"bitlen_stream": bitlen_stream_cases(ref, orc)}, f, separators=(",", ":"))
            f.write("\n")
        print("wrote", path, os.path.getsize(path), "bytes")
        return
    out = {}

    # (i) fresh ciphertexts ---------------------------------------------------------
    enc = []
    for (n, d) in [(1247, 16), (4096, 32), (63, 4), (64, 4), (65, 4), (100, 1)]:
        for seed in (1, 20260101):
            key = make_key(n, d, seed)
            bits = [1, 0, 0, 1, 0]
            ct, _ = ref.encrypt_seq(n, d, key, seed, bits)
            dl = (n + 63) // 64
            dec = [ref.decrypt(n, d, key, ct[i * dl:(i + 1) * dl], canonical_bitlen(n, 1))
                   for i in range(len(bits))]
            enc.append(dict(n=n, d=d, seed=seed, key=ints(key), bits=bits, ct=hexs(ct), dec=dec))
    out["encrypt"] = enc

    # (ii) multiply KATs --------------------------------------------------------------
    mul = []
    for (n, d) in [(1247, 16), (65, 4), (64, 4)]:
        dl = (n + 63) // 64
        for (t1, t2) in [(1, 1), (1, 2), (2, 1), (2, 2), (3, 5)]:
            a = orc.synth(0xA0 + t1, n, 0, t1 * dl)
            b = orc.synth(0xB0 + t2, n, 0, t2 * dl)
            bl1 = (np.arange(a.size, dtype=np.uint64) % 60) + 1   # non-canonical on purpose:
            bl2 = (np.arange(b.size, dtype=np.uint64) % 50) + 5   # shows the left-bitlen rule
            v, bl = ref.mul(n, d, a, bl1, b, bl2)
            mul.append(dict(n=n, d=d, t1=t1, t2=t2, a=hexs(a), b=hexs(b), bitlen_a=ints(bl1),
                            bitlen_b=ints(bl2), out=hexs(v), bitlen_out=ints(bl)))
    out["mul"] = mul

    # (iii) add KATs ----------------------------------------------------------------
    add = []
    for (n, d) in [(1247, 16), (65, 4)]:
        dl = (n + 63) // 64
        for (t1, t2) in [(1, 1), (2, 3), (4, 1)]:
            a = orc.synth(0xC0 + t1, n, 0, t1 * dl)
            b = orc.synth(0xD0 + t2, n, 0, t2 * dl)
            bl1, bl2 = canonical_bitlen(n, t1), canonical_bitlen(n, t2)
            v, bl = ref.add(n, d, a, bl1, b, bl2)
            add.append(dict(n=n, d=d, t1=t1, t2=t2, a=hexs(a), b=hexs(b), out=hexs(v),
                            bitlen_out=ints(bl)))
    out["add"] = add

    # (iv) multi-term decrypt ----------------------------------------------------------
    dec = []
    for (n, d) in [(1247, 16), (4096, 32), (65, 4)]:
        dl = (n + 63) // 64
        key = make_key(n, d, 77)
        mask = orc.key_mask(n, key)
        for terms, hits in [(1, 1), (1, 0), (2, 2), (5, 3), (40, 17)]:
            v = orc.synth(0xE0 + terms, n, 0, terms * dl).reshape(terms, dl)
            v[:hits] |= mask
            w, b = int(key[0]) // 64, 63 - int(key[0]) % 64
            v[hits:, w] &= ~np.uint64(1 << b)
            flat = np.ascontiguousarray(v.reshape(-1))
            bit = ref.decrypt(n, d, key, flat, canonical_bitlen(n, terms))
            dec.append(dict(n=n, d=d, key=ints(key), terms=terms, seed=0xE0 + terms, hits=hits,
                            v=hexs(flat) if terms <= 5 else None, bit=bit,
                            digest="%016x" % orc.digest(flat)))
    out["decrypt"] = dec

    # (v) permutations -----------------------------------------------------------------
    perm = []
    for (n, d) in [(1247, 16), (65, 4), (63, 4)]:
        dl = (n + 63) // 64
        seed = 31337
        key = make_key(n, d, seed)
        p = ref.perm_random(n, seed)
        inv = ref.perm_inverse(p)
        comp = ref.perm_compose(p, inv)
        cts, _ = ref.encrypt_seq(n, d, key, seed, [1, 0, 1])
        pkey = ref.permute_key(n, d, p, key)
        single, _ = ref.permute_ciphertext(n, d, p, cts[:dl], canonical_bitlen(n, 1))
        multi, _ = ref.permute_ciphertext(n, d, p, cts, canonical_bitlen(n, 3))
        perm.append(dict(n=n, d=d, seed=seed, key=ints(key), perm=ints(p), inverse=ints(inv),
                         compose_is_identity=bool(np.array_equal(comp, np.arange(n, dtype=np.uint64))),
                         ct=hexs(cts), permuted_key=ints(pkey), permuted_first=hexs(single),
                         permuted_multi=hexs(multi),
                         dec_permuted=ref.decrypt(n, d, pkey, single, canonical_bitlen(n, 1))))
    out["permutation"] = perm

    # (vi) digests of large products -----------------------------------------------------
    dig = []
    n, d = 1247, 16
    dl = 20
    for t in (32, 256, 1024):
        a = orc.synth(0x43534743 + t, n, 0, t * dl)
        b = orc.synth(0x43534743 + 7 * t, n, 0, t * dl)
        bl = canonical_bitlen(n, t)
        v, _ = ref.mul(n, d, a, bl, b, bl)
        dig.append(dict(n=n, d=d, t1=t, t2=t, seed_a=0x43534743 + t, seed_b=0x43534743 + 7 * t,
                        out_len=int(v.size), digest="%016x" % orc.digest(v),
                        first_words=hexs(v[:4]), last_words=hexs(v[-4:])))
    out["digest"] = dig

    # (vii) keygen: seed recovered from the clock window --------------------------------
    kg = []
    for (n, d) in [(1247, 16), (4096, 32)]:
        key, t0, t1 = ref.keygen(n, d)
        for t in range(t0 - 1, t1 + 2):
            cand, used = orc.keygen(n, d, glibc_draws(t, 64 * d + 64))
            if np.array_equal(cand, key):
                kg.append(dict(n=n, d=d, srand=t, key=ints(key), draws_used=used))
                break
        else:
            raise SystemExit("could not recover keygen seed")
    out["keygen"] = kg

    # (viii) the reference's own demo flow (tests/basic_operations.cpp) -------------------
    n, d, seed = 1247, 16, 424242
    key = make_key(n, d, seed)
    cts, _ = ref.encrypt_seq(n, d, key, seed, [1, 0])
    bl = canonical_bitlen(n, 1)
    added, abl = ref.add(n, d, cts[:20], bl, cts[20:], bl)
    mult, mbl = ref.mul(n, d, cts[:20], bl, cts[20:], bl)
    out["basic_operations"] = dict(
        n=n, d=d, seed=seed, key=ints(key), c1=hexs(cts[:20]), c0=hexs(cts[20:]),
        added=hexs(added), multiplied=hexs(mult),
        dec_added=ref.decrypt(n, d, key, added, abl), dec_multiplied=ref.decrypt(n, d, key, mult, mbl))

    # (ix) text forms (operator<<) of every class, small context so the strings stay short ----
    n, d, seed = 130, 5, 77
    key = make_key(n, d, seed)
    cts, _ = ref.encrypt_seq(n, d, key, seed, [1, 0])
    perm = ref.perm_random(n, seed)
    out["text"] = dict(
        n=n, d=d, seed=seed, key=ints(key), ct=hexs(cts), perm=ints(perm),
        ciphertext=ref.text("ciphertext", n, d, cts, canonical_bitlen(n, 2), cts.size),
        key_text=ref.text("key", n, d, key, None, d),
        context=ref.text("context", n, d),
        plaintext=[ref.text("plaintext", n, d, length=0), ref.text("plaintext", n, d, length=1)],
        permutation=ref.text("permutation", n, d, perm, None, n))

    path = os.path.join(HERE, "csgn_kat.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")
    path = os.path.join(HERE, "csgn_kat_bitlen.json")
    with open(path, "w") as f:
        json.dump({"bitlen_stream": bitlen_stream_cases(ref, orc)}, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
