import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the hyphen-named package directory vloam-cmu-16833_amd as module `vloam_amd`."""
    if "vloam_amd" in sys.modules:
        return sys.modules["vloam_amd"]
    path = os.path.join(ROOT, "vloam-cmu-16833_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("vloam_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["vloam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_synth():
    load_pkg()
    return importlib.import_module("vloam_amd.synth")


@pytest.fixture(scope="session")
def vl():
    return load_pkg()


@pytest.fixture(scope="session")
def synth():
    return load_synth()


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.build()
    return _orc


_seq_cache = {}


@pytest.fixture(scope="session")
def sweeps(synth):
    """Cached synthetic sweeps: sweeps(n_rings, n_azimuth, k) -> float32 [n, 4]."""
    def get(n_rings, n_az, k, n_sweeps=40):
        key = (n_rings, n_az, n_sweeps)
        if key not in _seq_cache:
            _seq_cache[key] = (synth.SynthSequence(n_rings=n_rings, n_azimuth=n_az, n_sweeps=n_sweeps), {})
        seq, cache = _seq_cache[key]
        if k not in cache:
            cache[k] = seq.sweep(k)
        return cache[k]
    return get
