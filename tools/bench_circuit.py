#!/usr/bin/env python3
"""BASELINE config 5 (depth-16 add/mul circuit on single ciphertexts, Context(4096,32)) through the
public class API: the genuine reference on one host core (oracle/_ref) beside the drop-in on the
GPU (tests/cpp/libdropin_refdriver.so = the same driver source linked against libcertFHE.so).
Dev tool; the numbers are quoted in DESIGN.md section 5."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.binding import Ref, REF_SO

LIBDIR = os.path.join(ROOT, "csgn_amd", "lib")
DRIVER_SO = os.path.join(ROOT, "tests", "cpp", "libdropin_refdriver.so")
src = os.path.join(ROOT, "oracle", "ref_driver.cpp")
if not os.path.exists(DRIVER_SO) or os.path.getmtime(DRIVER_SO) < os.path.getmtime(src):
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include", "certfhe"),
                           "-I" + os.path.join(ROOT, "include"), "-o", DRIVER_SO, src, "-L" + LIBDIR, "-lcertFHE",
                           "-lcsgn_hip", "-Wl,-rpath," + LIBDIR])
libs = [("drop-in (GPU)", Ref(DRIVER_SO))]
if os.path.exists(REF_SO):
    libs.append(("reference (1 host core)", Ref(REF_SO)))
for n, d, levels in [(4096, 32, 16), (1247, 16, 16), (1247, 16, 20)]:
    for name, lib in libs:
        lib.time_circuit(n, d, levels, 3)                       # warm-up
        iters = 200
        t, terms, bit = lib.time_circuit(n, d, levels, iters)
        print(f"Context({n},{d}) depth {levels}: {name:<24} {t / iters * 1e6:9.1f} us per circuit "
              f"({terms} terms, bit {bit})", flush=True)
