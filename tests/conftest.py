import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")

# The library honours its LOL_GPU_* A/B switches only beside LOL_GPU_TUNING=1 (include/lol_gpu.h, lol_gpu_tuning_switches): the
# tests that compare code paths set such switches, so the test session opts in.  tests/test_cabi.py checks the fence itself in
# processes that do not.
os.environ.setdefault("LOL_GPU_TUNING", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the in-tree libraries exist (the prebuilt .so files travel with the repo snapshot)."""
    need = [os.path.join(ROOT, "loltracer_amd", "lib", "liblol_scene.so"),
            os.path.join(ROOT, "loltracer_amd", "lib", "liblol_gpu.so"),
            os.path.join(ROOT, "loltracer_amd", "lib", "lol_headless"),
            os.path.join(ROOT, "oracle", "liblol_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__ as g
        g.build()


def scene_path(name: str) -> str:
    return os.path.join(SCENES, name + ".lol")


@pytest.fixture(scope="session")
def scenes():
    from loltracer_amd import scene as S
    return {n: S.Scene.parse_file(scene_path(n)) for n in ("scene", "scene2", "scene3", "scene4")}
