import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


def scene_from_golden(g):
    from mve_amd.scene_io import Camera, Feature, SceneData
    cams = []
    for c in g["cams"]:
        cams.append(Camera(flen=float(c[0]), paspect=float(c[1]), ppoint=[float(c[2]), float(c[3])],
                           rot=[float(v) for v in c[4:13]], trans=[float(v) for v in c[13:16]]))
    imgs = [np.ascontiguousarray(im) for im in g["imgs"]]
    off, ref = g["foff"], g["fref"]
    feats = [Feature([float(v) for v in g["fpos"][i]], [int(v) for v in ref[off[i]:off[i + 1]]])
             for i in range(len(g["fpos"]))]
    return SceneData(cams, imgs, feats)


@pytest.fixture(scope="session")
def g1():
    return dict(np.load(os.path.join(GOLDEN, "g1_5views_160x120.npz")))


@pytest.fixture(scope="session")
def g1b():
    return dict(np.load(os.path.join(GOLDEN, "g1b_5views_322x241_scale1.npz")))


@pytest.fixture(scope="session")
def g2():
    return dict(np.load(os.path.join(GOLDEN, "g2_2views_160x120_k1.npz")))


@pytest.fixture(scope="session")
def g1_fw():
    return dict(np.load(os.path.join(GOLDEN, "g1_filter_widths.npz")))


@pytest.fixture(scope="session")
def h1():
    return dict(np.load(os.path.join(GOLDEN, "h1_hard_9views_208x156.npz")))


@pytest.fixture(scope="session")
def h1_scene(h1):
    return scene_from_golden(h1)


@pytest.fixture(scope="session")
def w1():
    """Scene W1 (42 views): the reference with 40 global views / six and eight local views (make_golden_wide.py)."""
    return dict(np.load(os.path.join(GOLDEN, "w1_wide_42views_112x84.npz")))


@pytest.fixture(scope="session")
def w1_scene(w1):
    return scene_from_golden(w1)


@pytest.fixture(scope="session")
def w2():
    """Scene W2 (100 views): the reference with 80 global views / ten and sixteen local views (make_golden_wide2.py)."""
    return dict(np.load(os.path.join(GOLDEN, "w2_wider_100views_96x72.npz")))


@pytest.fixture(scope="session")
def w2_scene(w2):
    return scene_from_golden(w2)


@pytest.fixture(scope="session")
def g1_scene(g1):
    return scene_from_golden(g1)


@pytest.fixture(scope="session")
def g1b_scene(g1b):
    return scene_from_golden(g1b)


@pytest.fixture(scope="session")
def g2_scene(g2):
    return scene_from_golden(g2)


@pytest.fixture(autouse=True)
def _no_fault_injection_left_behind(request):
    """A GPU test that fails half-way must not leave its injected fault (mi_dmrecon_debug_inject_footprint) to the tests after it."""
    yield
    if "gpu" in request.keywords:
        from mve_amd import api
        api.debug_inject_footprint(-1)


@pytest.fixture(scope="session")
def gpu_ctx():
    """One HIP context for the whole GPU test session (fails loudly without the library / a GPU)."""
    from mve_amd import api
    assert api.device_count() > 0, "no HIP device visible: -m gpu tests need an MI355X"
    return api.Context(0)


def map_parity(a_depth, a_conf, b_depth, b_conf):
    """Map-level parity metrics between two reconstructions of the same view."""
    ma, mb = a_depth > 0, b_depth > 0
    both = ma & mb
    iou = both.sum() / max((ma | mb).sum(), 1)
    rel = np.abs(a_depth[both] - b_depth[both]) / b_depth[both]
    cd = np.abs(a_conf[both] - b_conf[both])
    m = dict(iou=float(iou), rel_med=float(np.median(rel)), rel_p99=float(np.percentile(rel, 99)),
             conf_med=float(np.median(cd)), conf_p99=float(np.percentile(cd, 99)),
             n_a=int(ma.sum()), n_b=int(mb.sum()))
    log = os.environ.get("MI_TEST_LOG_METRICS")          # (a file to append every measured metric set to: how far the bounds are)
    if log:
        import json
        with open(log, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), **m}) + "\n")
    return m
