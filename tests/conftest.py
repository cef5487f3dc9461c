import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")

# The library honours its LOL_GPU_* A/B switches only beside LOL_GPU_TUNING=1 (include/lol_gpu.h, lol_gpu_tuning_switches).  The test
# session does NOT opt in (rounds 4 - 5 did, and every test ran in a process state no product host has): a test that compares
# code paths sets LOL_GPU_TUNING=1 with monkeypatch beside the switch it sets, and the fixture below sees to it that no other does.
os.environ.pop("LOL_GPU_TUNING", None)


@pytest.fixture(autouse=True)
def _fence_closed_unless_asked():
    assert "LOL_GPU_TUNING" not in os.environ, "a test left LOL_GPU_TUNING set (use monkeypatch: it is undone after the test)"
    yield


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
