"""dev: how often does a circuit whose long-uniform decrypt zero-fills through a MEMSET node (knob zero_memset=1)
return a wrong bit, against the same circuit with the zero fill as a kernel node?  (gpurun_out/s3: one wrong bit in
the second of five runs with the memset node.)  120 runs per form, results against the clear circuit."""
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from csgn_amd.batch import HipPath
from csgn_amd import capi
from csgn_amd.capi import check
hip = HipPath(0); lib = hip.lib
n, d, batch, t = 1247, 16, 3, 70
dl = hip.default_len(n)
key = np.random.default_rng(3).permutation(n)[:d].astype(np.uint64)
dmask = hip.upload(hip.key_mask(n, key)); dkey = hip.upload(key)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for graph_knob in (1, 0, 1, 0):
    capi.set_tuning("zero_memset", graph_knob)
    c = C.c_void_p(); check(lib.csgn_circuit_create(n, batch, C.byref(c)))
    def new(fn, *a):
        v = C.c_uint32(); check(fn(c, *a, C.byref(v))); return v.value
    a, b = new(lib.csgn_circuit_input, t), new(lib.csgn_circuit_input, t)
    p = new(lib.csgn_circuit_mul, a, b)
    bid = new(lib.csgn_circuit_decrypt, p, dmask.data_ptr())
    check(lib.csgn_circuit_build(c))
    capi.set_tuning("zero_memset", 0)
    wrong, wrong_direct, first = 0, 0, None
    for rnd in range(runs):
        plain = np.random.default_rng(rnd).integers(0, 2, size=2 * t * batch).astype(np.uint8)
        fresh = hip.encrypt_device_rng(n, d, hip.upload(plain), dkey, dmask, seed=rnd + 40)
        half = batch * t * dl
        check(lib.csgn_memcpy_d2d(lib.csgn_circuit_value(c, a), fresh.data_ptr(), half * 8, hip.stream))
        check(lib.csgn_memcpy_d2d(lib.csgn_circuit_value(c, b), fresh[half:].data_ptr(), half * 8, hip.stream))
        check(lib.csgn_circuit_run(c, hip.stream))
        gb = torch.empty(batch, dtype=torch.uint8, device=fresh.device)
        check(lib.csgn_memcpy_d2d(gb.data_ptr(), lib.csgn_circuit_bits(c, bid), batch, hip.stream))
        prod = hip.mul_uniform(n, batch, t, t, fresh[:half], fresh[half:])
        capi.set_tuning("zero_memset", graph_knob)          # the direct call zero-fills the same way as the graph
        direct = hip.download(hip.decrypt_uniform(n, batch, t * t, prod, dmask)).tolist()
        capi.set_tuning("zero_memset", 0)
        pb = plain.reshape(2, batch, t)
        clear = (np.bitwise_xor.reduce(pb[0], axis=1) & np.bitwise_xor.reduce(pb[1], axis=1)).tolist()
        got = hip.download(gb).tolist()
        if got != clear:
            wrong += 1
            first = first or (rnd, got, clear)
        wrong_direct += direct != clear
    print(f"zero fill as {'MEMSET node / hipMemsetAsync' if graph_knob else 'kernel':<30}: graph wrong in {wrong} of {runs} runs"
          f"{' (first: run %d, got %s, clear %s)' % first if first else ''}; direct call wrong in {wrong_direct}", flush=True)
    lib.csgn_circuit_destroy(c)
