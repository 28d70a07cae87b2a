"""certFHE::ShardGroup / ShardedBatch (include/certfhe/ShardedBatch.h, libcertFHE_shard.so): the
class-level face of the sharded batch, driven through
tests/cpp/sharded_driver.cpp.

CPU box: the library and the driver build, the library exports the class surface, and the driver
fails LOUDLY without a GPU (no CPU fallback).
GPU box (-m gpu): on every visible GPU (world 1 on the one-GPU box) the sharded batch holds exactly
the words certFHE::CiphertextBatch holds, for both forms of the gather; a failing rank surfaces as
an exception instead of a hang.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "sharded_driver.cpp")
DRIVER = os.path.join(ROOT, "tests", "cpp", "sharded_driver")
LIBDIR = os.path.join(ROOT, "csgn_amd", "lib")


@pytest.fixture(scope="module")
def driver():
    from csgn_amd import build
    build.build_all()
    deps = [SRC] + [os.path.join(LIBDIR, l) for l in ("libcertFHE_shard.so", "libcertFHE.so", "libcsgn_shard.so")]
    if not os.path.exists(DRIVER) or os.path.getmtime(DRIVER) < max(os.path.getmtime(d) for d in deps):
        inc = os.path.join(ROOT, "include")
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(inc, "certfhe"), "-I" + inc,
                               "-o", DRIVER, SRC, "-L" + LIBDIR, "-lcertFHE_shard", "-lcertFHE", "-lcsgn_shard",
                               "-lcsgn_hip", "-lpthread", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return DRIVER


def test_library_exports_the_class_surface(driver):
    out = subprocess.run(["nm", "-DC", "--defined-only", os.path.join(LIBDIR, "libcertFHE_shard.so")],
                         capture_output=True, text=True, check=True).stdout
    for sym in ("certFHE::ShardGroup::ShardGroup(", "certFHE::ShardGroup::collective", "certFHE::ShardGroup::injectFailure(",
                "certFHE::ShardedBatch::encrypt(", "certFHE::ShardedBatch::encryptProduct(", "certFHE::ShardedBatch::operator*(",
                "certFHE::ShardedBatch::operator+(", "certFHE::ShardedBatch::decrypt(", "certFHE::ShardedBatch::termCounts()",
                "certFHE::ShardedBatch::digest()", "certFHE::ShardedBatch::values(", "certFHE::ShardedBatch::applyPermutation(",
                "certFHE::ShardedBatch::decryptProduct(", "certFHE::ShardedBatch::decryptSum("):
        assert sym in out, sym
    # RCCL comes in through libcsgn_shard.so only; the single-GPU class library must stay free of it
    needed = lambda lib: " ".join(l for l in subprocess.run(["readelf", "-d", os.path.join(LIBDIR, lib)], capture_output=True,
                                                              text=True, check=True).stdout.splitlines() if "NEEDED" in l)
    assert "libcsgn_shard.so" in needed("libcertFHE_shard.so")
    assert "rccl" not in needed("libcertFHE.so") and "csgn_shard" not in needed("libcertFHE.so")


def test_fails_loudly_without_gpu(driver):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([driver, "compare", "16"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and ("no HIP device" in p.stderr or "no CPU fallback" in p.stderr or "no ROCm-capable" in p.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("count,n,d", [(1000, 1247, 16), (70001, 1247, 16), (257, 4096, 32), (5, 64, 2)])
def test_sharded_batch_equals_the_one_gpu_batch(driver, count, n, d):
    p = subprocess.run([driver, "compare", str(count), str(n), str(d)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "FAIL" not in p.stdout and p.stdout.count("OK ") >= 28
    assert "grouped ncclBroadcast" in p.stdout and "RCCL 2." in p.stdout


@pytest.mark.gpu
def test_a_failing_rank_is_an_exception_not_a_hang(driver):
    p = subprocess.run([driver, "fail", "4096"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "FAIL" not in p.stdout and "injected failure" in p.stdout
