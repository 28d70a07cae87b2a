"""pytest configuration: markers + shared fixtures.

`-m "not gpu"` : oracle vs golden vectors / vs the compiled reference, host logic, C-ABI
                 symbol export, gloo world_size-2 sharding.  Runs without a GPU.
`-m gpu`       : parity tests proper -- every call goes through the C-ABI of
                 csgn_amd/lib/libcsgn_hip.so on cuda:0 and is checked against the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The genuine reference (oracle/_ref).  Present in the dev container and, as a prebuilt
    .so, on the GPU box; tests that need it skip elsewhere."""
    from oracle.binding import load_ref
    r = load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libcsgn_ref.so not built (reference sources absent)")
    return r


class _Knobs:
    """Tuning knobs of libcsgn_hip.so for one test (csgn_set_tuning): set()/unset() take the knob
    name ('mul_flat') or its environment spelling ('CSGN_MUL_FLAT'); everything is restored at
    teardown.  The library never reads the environment after it is loaded."""

    def __init__(self):
        from csgn_amd import capi
        self.capi = capi
        self.saved = {n: capi.get_tuning(n) for n in capi.tuning_names()}

    def set(self, name, value):
        self.capi.set_tuning(name, int(value))

    def unset(self, name):
        key = name[5:].lower() if name.upper().startswith("CSGN_") else name
        self.capi.set_tuning(key, self.saved[key])

    def restore(self):
        for n, v in self.saved.items():
            self.capi.set_tuning(n, v)


@pytest.fixture
def knobs():
    k = _Knobs()
    yield k
    k.restore()
