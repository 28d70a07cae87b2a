"""The circuit compiler (csgn_circuit_optimize / csgn_circuit_build, csgn_amd/csrc/csgn_circuit.hip) on the HOST:
csgn_circuit_plan_json runs its passes -- decrypt fusion, dead-node removal, placement of producers into the sums that
consume them, liveness-driven layout of the block -- without a device, and this file EXECUTES the resulting plan in
numpy (every kept node reads and writes one byte array at the plan's offsets and pitches) against the same circuit
evaluated value by value with the oracle.  A wrong offset, an overlapping pair of live regions, a fusion that fired on a
shared sub-expression or a node dropped while something still read it all show up as wrong words or bits here, before
any GPU is involved.  The GPU tests (tests/test_gpu_parity.py::test_circuit_compiled_*) repeat the property on the
real kernels.  Reference semantics: src/Ciphertext.cpp:107-122 (add = concatenation), :146-163 (all-pairs AND),
src/SecretKey.cpp:104-147 (decrypt = XOR over terms of AND over key positions)."""
import ctypes as C
import json

import numpy as np
import pytest

from csgn_amd import capi
from csgn_amd.capi import check

REUSE, PLACE, FUSE, PUSHDOWN, HOIST = 1, 2, 4, 8, 16
CORE, ALL = 7, 23
FOREVER = 2 ** 31 - 1
FAKE_PTR = 0x1000                                   # a device pointer the host passes never read


@pytest.fixture(scope="module")
def lib():
    return capi.load_library()


class Described:
    """A circuit described through the C ABI and, beside it, the same description as Python tuples."""

    def __init__(self, lib, n, batch, mask_ptr=FAKE_PTR):
        self.lib, self.n, self.batch, self.mask_ptr = lib, n, batch, mask_ptr
        self.dl = int(lib.csgn_default_len(n))
        self.c = C.c_void_p()
        check(lib.csgn_circuit_create(n, batch, C.byref(self.c)))
        self.terms = []                             # per value
        self.nodes = []                             # ("in",) / ("add", a, b) / ("mul", a, b) per value
        self.decrypts = []                          # value ids, in bits_id order
        self.outputs = set()

    def close(self):
        self.lib.csgn_circuit_destroy(self.c)

    def _new(self, fn, *args):
        v = C.c_uint32()
        check(fn(self.c, *args, C.byref(v)))
        return v.value

    def input(self, terms=1):
        v = self._new(self.lib.csgn_circuit_input, terms)
        self.terms.append(terms)
        self.nodes.append(("in",))
        return v

    def add(self, a, b):
        v = self._new(self.lib.csgn_circuit_add, a, b)
        self.terms.append(self.terms[a] + self.terms[b])
        self.nodes.append(("add", a, b))
        return v

    def mul(self, a, b):
        v = self._new(self.lib.csgn_circuit_mul, a, b)
        self.terms.append(self.terms[a] * self.terms[b])
        self.nodes.append(("mul", a, b))
        return v

    def decrypt(self, a):
        bid = self._new(self.lib.csgn_circuit_decrypt, a, self.mask_ptr)
        assert bid == len(self.decrypts)
        self.decrypts.append(a)
        return bid

    def output(self, v):
        check(self.lib.csgn_circuit_output(self.c, v))
        self.outputs.add(v)

    def plan(self, flags):
        check(self.lib.csgn_circuit_optimize(self.c, flags))
        buf = C.create_string_buffer(1 << 22)
        check(self.lib.csgn_circuit_plan_json(self.c, buf, len(buf)))
        return json.loads(buf.value.decode())


def evaluate(d, inputs):
    """Every value of the description, element by element: [batch, terms, dl] uint64 arrays."""
    vals = []
    for node in d.nodes:
        if node[0] == "in":
            vals.append(inputs[len(vals)])
        elif node[0] == "add":
            vals.append(np.concatenate([vals[node[1]], vals[node[2]]], axis=1))
        else:
            a, b = vals[node[1]], vals[node[2]]
            vals.append((a[:, :, None, :] & b[:, None, :, :]).reshape(d.batch, -1, d.dl))
    return vals


def decrypt_bits(vals, mask):
    hits = np.all((vals & mask) == mask, axis=2)               # [batch, terms]
    return (hits.sum(axis=1) & 1).astype(np.uint8)


def execute(d, plan, inputs, mask):
    """Run the PLAN: kept nodes only, in order, every read and write at the plan's byte offsets and pitches."""
    block = np.zeros(plan["bytes"] // 8 + 8, dtype=np.uint64)
    rng = np.random.default_rng(5)
    block[:] = rng.integers(0, 2 ** 63, size=block.size, dtype=np.uint64)       # stale bytes must never matter
    dl, B = d.dl, d.batch

    def view(v):
        pv = plan["values"][v]
        base, pitch, t = pv["offset"] // 8, pv["pitch"], d.terms[v]
        assert pv["offset"] % 8 == 0 and pitch >= t * dl
        idx = base + np.arange(B)[:, None] * pitch + np.arange(t * dl)[None, :]
        return idx

    def read(v):
        pv = plan["values"][v]
        assert pv["region"] and pv["parent"] < 0 and pv["pitch"] == d.terms[v] * dl, f"value {v} is read but not dense"
        return block[view(v)].reshape(B, d.terms[v], dl)

    def write(v, words):
        block[view(v)] = words.reshape(B, -1)

    for v, node in enumerate(d.nodes):
        if node[0] == "in":
            assert plan["values"][v]["addressable"]
            write(v, inputs[v])
    def copy_operand(op, side):
        """operand `side` of add `op` into its slice of the sum (what the add's own copy, or the prologue launch, does)"""
        ta = d.terms[op["a"]]
        out = plan["values"][op["out"]]
        base, pitch = out["offset"] // 8, out["pitch"]
        v = op["b"] if side else op["a"]
        idx = base + (ta * dl if side else 0) + np.arange(B)[:, None] * pitch + np.arange(d.terms[v] * dl)[None, :]
        block[idx] = read(v).reshape(B, -1)

    # the prologue: every hoisted copy (its source is an input) before the first node
    for op in plan["ops"]:
        if op["kind"] == 0 and not op["elided"]:
            for side, key in ((0, "hoist_a"), (1, "hoist_b")):
                if op[key]:
                    assert d.nodes[op["b"] if side else op["a"]][0] == "in" and not op["placed_b" if side else "placed_a"]
                    copy_operand(op, side)
    bits = {}
    for i, op in enumerate(plan["ops"]):
        if op["elided"]:
            continue
        if op["kind"] == 1:
            a, b = read(op["a"]), read(op["b"])
            write(op["out"], (a[:, :, None, :] & b[:, None, :, :]).reshape(B, -1, dl))
        elif op["kind"] == 0:
            ta = d.terms[op["a"]]
            out = plan["values"][op["out"]]
            base, pitch = out["offset"] // 8, out["pitch"]
            if op["placed_a"]:
                pa = plan["values"][op["a"]]
                assert pa["parent"] == op["out"] and pa["offset"] == out["offset"] and pa["pitch"] == pitch
            elif not op["hoist_a"]:
                copy_operand(op, 0)
            if not op["placed_b"] and not op["hoist_b"]:
                copy_operand(op, 1)
            elif op["placed_b"]:
                pb = plan["values"][op["b"]]
                assert pb["parent"] == op["out"] and pb["offset"] == out["offset"] + ta * dl * 8 and pb["pitch"] == pitch
        elif op["kind"] == 2:
            def ev(ei):
                e = plan["exprs"][ei]
                if e["kind"] < 0:
                    return decrypt_bits(read(e["value"]), mask)
                l, r = ev(e["l"]), ev(e["r"])
                return (l & r) if e["kind"] == 1 else (l ^ r)
            bid = sum(1 for o in plan["ops"][:i] if o["kind"] == 2)
            bits[bid] = ev(op["expr"]) if op["expr"] >= 0 else decrypt_bits(read(op["a"]), mask)
    return block, bits, read


def check_regions(plan):
    """No two regions that are live at the same node overlap."""
    regs = plan["regions"]
    for i, a in enumerate(regs):
        assert a["at"] % 256 == 0 and a["at"] + a["bytes"] <= plan["bytes"]
        for b in regs[i + 1:]:
            if a["from"] <= b["to"] and b["from"] <= a["to"]:
                assert a["at"] + a["bytes"] <= b["at"] or b["at"] + b["bytes"] <= a["at"], (a, b)


def random_circuit(lib, seed, n, batch, mask_ptr=FAKE_PTR, max_terms=400):
    rng = np.random.default_rng(seed)
    d = Described(lib, n, batch, mask_ptr)
    for _ in range(int(rng.integers(2, 6))):
        d.input(int(rng.integers(1, 4)))
    for _ in range(int(rng.integers(3, 14))):
        k = len(d.terms)
        # mostly chains (the newest value and something else), sometimes two old values: shared sub-expressions
        a = k - 1 if rng.random() < 0.6 else int(rng.integers(0, k))
        b = int(rng.integers(0, k))
        if rng.random() < 0.45 and d.terms[a] * d.terms[b] <= max_terms:
            d.mul(a, b)
        elif d.terms[a] + d.terms[b] <= max_terms:
            d.add(a, b)
    k = len(d.terms)
    for v in sorted(set(int(x) for x in rng.integers(0, k, size=int(rng.integers(1, 4))))):
        d.decrypt(v)
    if rng.random() < 0.5:
        d.decrypt(k - 1)
    for v in sorted(set(int(x) for x in rng.integers(0, k, size=int(rng.integers(0, 3))))):
        d.output(v)
    return d


@pytest.mark.parametrize("flags", [ALL, CORE, REUSE, PLACE, FUSE, HOIST, REUSE | HOIST, PLACE | FUSE, ALL | PUSHDOWN, REUSE | PUSHDOWN])
def test_compiled_plans_of_random_circuits_compute_what_the_tape_computes(lib, flags):
    """60 random DAGs per flag set (chains, shared sub-expressions, squares a*a and doubles a+a, values nobody reads,
    several decrypts, random retained outputs): the executed plan's retained words and all bits equal the direct
    evaluation; no two live regions overlap; nothing that has two readers was placed or dissolved."""
    for seed in range(60):
        n, batch = ((130, 3), (128, 2), (64, 4))[seed % 3]
        d = random_circuit(lib, 1000 + seed, n, batch)
        try:
            plan = d.plan(flags)
            check_regions(plan)
            rng = np.random.default_rng(seed)
            inputs = {v: rng.integers(0, 2 ** 63, size=(batch, d.terms[v], d.dl), dtype=np.uint64) | np.uint64(1 << 63)
                      for v, node in enumerate(d.nodes) if node[0] == "in"}
            mask = np.zeros(d.dl, dtype=np.uint64)
            mask[0] = np.uint64((1 << 63) | (1 << 62))        # hit iff two bits set: products of random words hit 1 in 2**k
            want = evaluate(d, inputs)
            block, bits, read = execute(d, plan, inputs, mask)
            for bid, v in enumerate(d.decrypts):
                assert np.array_equal(bits[bid], decrypt_bits(want[v], mask)), (seed, bid)
            for v in d.outputs | {i for i, nd in enumerate(d.nodes) if nd[0] == "in"}:
                assert plan["values"][v]["addressable"], (seed, v)
                assert np.array_equal(read(v), want[v]), (seed, v)
            # readers per value among the kept nodes and the decrypt leaves
            readers = [0] * len(d.terms)
            for op in plan["ops"]:
                if op["elided"] or (op["kind"] == 2 and op["expr"] >= 0):
                    continue
                readers[op["a"]] += 1
                if op["kind"] in (0, 1):
                    readers[op["b"]] += 1
            for e in plan["exprs"]:
                if e["kind"] < 0:
                    readers[e["value"]] += 1
            for v, pv in enumerate(plan["values"]):
                if pv["parent"] >= 0:
                    assert readers[v] == 1 and v not in d.outputs and d.nodes[v][0] != "in", (seed, v)
                if not pv["region"] and pv["parent"] < 0:
                    assert readers[v] == 0, (seed, v)
            if not flags & PLACE:
                assert all(pv["parent"] < 0 for pv in plan["values"])
            if not flags & HOIST:
                assert not any(op["hoist_a"] or op["hoist_b"] for op in plan["ops"])
            if not flags & (FUSE | PUSHDOWN):
                assert not plan["exprs"]
        finally:
            d.close()


def test_tape_mode_keeps_every_value_in_a_region_of_its_own(lib):
    d = random_circuit(lib, 7, 130, 3)
    try:
        plan = d.plan(0)
        assert all(pv["region"] and pv["addressable"] and pv["parent"] < 0 for pv in plan["values"])
        assert not any(op["elided"] or op["placed_a"] or op["placed_b"] or op["hoist_a"] or op["hoist_b"] for op in plan["ops"])
        check_regions(plan)
        assert all(r["to"] == FOREVER or r["from"] == r["to"] for r in plan["regions"])
    finally:
        d.close()


def config5(lib, n, batch, levels=16, mask_ptr=FAKE_PTR):
    d = Described(lib, n, batch, mask_ptr)
    ins = [d.input(1) for _ in range(1 + levels // 2 + 2 * (levels // 2))]
    x, k = ins[0], 1
    for level in range(1, levels + 1):
        if level % 2:
            x = d.add(x, ins[k]); k += 1
        else:
            x = d.mul(x, d.add(ins[k], ins[k + 1])); k += 2
    d.decrypt(x)
    return d, x


def test_config5_plan_elides_the_last_product_and_every_copy_of_a_product(lib):
    """BASELINE config 5 (x <- x + e / x <- x * (e + e'), depth 16, 766 terms, then Dec): compiled, the last product is
    never written (Dec(x15 * r) = Dec(x15) & Dec(r)), every other product is written straight into the sum that
    consumes it, and the block holds the peak live set instead of all 41 values."""
    d, x = config5(lib, 4096, 8)
    try:
        tape = d.plan(0)
        comp = d.plan(ALL)
        check_regions(comp)
        muls = [op for op in comp["ops"] if op["kind"] == 1]
        assert len(muls) == 8 and [m["elided"] for m in muls] == [False] * 7 + [True]
        x_adds = [op for op in comp["ops"] if op["kind"] == 0 and d.nodes[op["a"]][0] == "mul"]
        assert len(x_adds) == 7 and all(op["placed_a"] and not op["placed_b"] and op["hoist_b"] for op in x_adds)
        # ... and every copy of an input (the other half of those adds, x0 + e, the eight sums e + e') is in the prologue:
        # no add launches a kernel of its own any more
        adds = [op for op in comp["ops"] if op["kind"] == 0 and not op["elided"]]
        assert all((op["placed_a"] or op["hoist_a"]) and (op["placed_b"] or op["hoist_b"]) for op in adds)
        assert sum(op["hoist_a"] + op["hoist_b"] for op in adds) == 25          # every one of the 25 inputs, once
        dec = [op for op in comp["ops"] if op["kind"] == 2][0]
        root = comp["exprs"][dec["expr"]]
        assert root["kind"] == 1 and comp["exprs"][root["l"]]["kind"] < 0 and comp["exprs"][root["r"]]["kind"] < 0
        assert not comp["values"][x]["region"] and tape["values"][x]["region"]
        # the tape holds every value (1 + 2 + 2 + 4 + ... + 766 terms and the inputs); the compiled block the inputs,
        # the two largest neighbouring sums and scratch
        assert comp["bytes"] * 2 < tape["bytes"]
        # kept as an output, the last product is computed after all
        d.output(x)
        kept = d.plan(ALL)
        assert kept["values"][x]["region"] and kept["values"][x]["addressable"]
        assert not any(op["elided"] for op in kept["ops"])
        # with PUSHDOWN the decrypt reaches the 25 inputs
        d2, x2 = config5(lib, 4096, 8)
        try:
            deep = d2.plan(ALL | PUSHDOWN)
            leaves = [e for e in deep["exprs"] if e["kind"] < 0]
            assert len(leaves) == 25 and all(d2.nodes[e["value"]][0] == "in" for e in leaves)
            assert all(op["elided"] for op in deep["ops"] if op["kind"] != 2)
        finally:
            d2.close()
    finally:
        d.close()


def test_optimize_and_output_argument_checks(lib):
    d = Described(lib, 130, 2)
    try:
        a = d.input(1)
        with pytest.raises(capi.CsgnError):
            check(lib.csgn_circuit_optimize(d.c, 64))                     # unknown flag
        with pytest.raises(capi.CsgnError):
            check(lib.csgn_circuit_output(d.c, 99))                       # no such value
        buf = C.create_string_buffer(64)
        with pytest.raises(capi.CsgnError):
            check(lib.csgn_circuit_plan_json(d.c, buf, len(buf)))         # no operations yet
        d.decrypt(d.add(a, a))
        with pytest.raises(capi.CsgnError):
            check(lib.csgn_circuit_plan_json(d.c, buf, len(buf)))         # buffer too small
        assert lib.csgn_circuit_block_bytes(d.c) == 0                     # not built
        stats = (C.c_uint64 * 8)()
        with pytest.raises(capi.CsgnError):
            check(lib.csgn_circuit_stats(d.c, stats))
    finally:
        d.close()
