"""GPU tests at BASELINE.json's full size (config 3: 20 views 1920x1080, scale 2 -> 480x270 maps).
The oracle cannot run 20 such views in seconds, so the full batch is checked through size-independent
properties (determinism, invariants of the maps, accuracy against the analytic ground truth of the
synthetic scene) and THREE views -- among them the two with the largest one-sided border strips -- are compared
with the CPU oracle under the map-level tolerance, the fill masks against the reference algorithm's own
order-sensitivity floor on the same views."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import map_parity
from mve_amd import api
from mve_amd.synth import CONFIGS, make_scene, true_depth

pytestmark = pytest.mark.gpu


_ORACLE_VIEW = """
import sys, numpy as np
sys.path.insert(0, %r)
from mve_amd.synth import CONFIGS, make_scene
from oracle import oracle as orc
cfg = CONFIGS["C3"]
S = orc.OracleScene(make_scene(cfg["params"]))
r = S.reconstruct(orc.make_settings(ref_view=int(sys.argv[1]), scale=cfg["scale"], local_neighbors=cfg["local_neighbors"]))
np.savez(sys.argv[2], d=r["depth"], c=r["conf"])
"""


@pytest.fixture(scope="module")
def c3_oracle(tmp_path_factory):
    """The CPU oracle on views 3, 8 and 12 of the C3 scene, and on 12 once more with its queue popped worst-first
    (ORC_QUEUE_ORDER=reverse): four subprocesses side by side, started
    before the GPU fixture so that they run while the GPU tests do.  Returns get(view, reverse=False) -> maps."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    td = tmp_path_factory.mktemp("c3_oracle")
    jobs = {}
    for v, rev in ((3, False), (8, False), (12, False), (12, True)):
        out = str(td / ("v%d%s.npz" % (v, "r" if rev else "")))
        env = dict(os.environ, OMP_NUM_THREADS="8")
        if rev:
            env["ORC_QUEUE_ORDER"] = "reverse"
        jobs[(v, rev)] = (subprocess.Popen([sys.executable, "-c", _ORACLE_VIEW % root, str(v), out], env=env), out)

    def get(view, reverse=False):
        p, out = jobs[(view, reverse)]
        assert p.wait(timeout=1200) == 0
        return np.load(out)
    yield get
    for p, _ in jobs.values():
        if p.poll() is None:
            p.kill()


@pytest.fixture(scope="module")
def c3(c3_oracle):
    cfg = CONFIGS["C3"]
    scene = make_scene(cfg["params"])
    ctx = api.Context(0)
    ctx.load_scene(scene)
    st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
    res = ctx.reconstruct(st, list(range(cfg["params"].n_views)), want_views=True)
    return cfg, scene, ctx, st, res, dict(ctx.last_stats)


def test_c3_invariants_and_accuracy(c3):
    cfg, scene, ctx, st, res, stats = c3
    p = cfg["params"]
    assert len(res) == 20 and res[0]["depth"].shape == (270, 480)
    fills, errs = [], []
    for v, r in enumerate(res):
        d, c = r["depth"], r["conf"]
        filled = c > 0
        assert (d[~filled] == 0).all() and (r["dz"][~filled] == 0).all() and (r["normal"][~filled] == 0).all()
        assert (d[filled] > 0).all() and c.max() <= 1.0
        assert not filled[:2].any() and not filled[-2:].any() and not filled[:, :2].any() and not filled[:, -2:].any()
        assert np.abs(np.linalg.norm(r["normal"][filled], axis=1) - 1).max() < 1e-4
        assert (r["views"][filled] >= 0).all() and (np.diff(r["views"][filled], axis=1) > 0).all()
        assert not (r["views"][filled] == v).any()                     # a view is never its own neighbour
        gt = true_depth(p, scene.cameras[v], 480, 270)
        fills.append(filled.mean())
        errs.append(np.median(np.abs(d[filled] - gt[filled])))
    assert np.mean(fills) > 0.85, fills
    assert np.median(errs) < 1.5e-2, errs                               # depth ~10, 480-px-wide level
    assert stats["n_filled"] == sum(int((r["conf"] > 0).sum()) for r in res)
    # work mix: the unit counts behind the roofline figure
    assert 25 < stats["n_eval"] / stats["n_patch"] < 40
    assert 1.0 < stats["n_patch"] / stats["n_filled"] < 2.5


def test_c3_global_view_selection_vs_oracle(c3):
    """The table-driven global view selection (scene-level parallax / visibility tables) picks exactly the
    reference's views for all 20 reference views of the full-size scene."""
    from oracle import oracle as orc
    cfg, scene, ctx, st, res, stats = c3
    S = orc.OracleScene(scene)
    for ref in range(cfg["params"].n_views):
        mine = ctx.global_view_selection(api.Settings(refViewNr=ref, scale=cfg["scale"]), ref)
        assert mine == S.global_vs(orc.make_settings(ref_view=ref, scale=cfg["scale"])), ref
    for ref in (0, 11):
        stn = api.Settings(refViewNr=ref, scale=cfg["scale"], globalVSMax=5)
        assert ctx.global_view_selection(stn, ref) == S.global_vs(orc.make_settings(ref_view=ref, scale=cfg["scale"], global_max=5))


def test_c3_view_selection_on_device(c3, monkeypatch):
    """gvs_device.hip agrees with the host loop -- which the test above pins to the oracle -- for every reference view
    of the full-size scene (the fixture's 20-view call is below the size from which the device is the default)."""
    import os
    cfg, scene, ctx, st, res, stats = c3
    if "MI_DMRECON_GVS_DEVICE" not in os.environ:                      # (the suite is also run with the switch forced)
        assert stats["gvs_on_device"] == 0
    for ref in range(cfg["params"].n_views):
        for gmax in (20, 4):
            s = api.Settings(refViewNr=ref, scale=cfg["scale"], globalVSMax=gmax)
            monkeypatch.setenv("MI_DMRECON_GVS_DEVICE", "0")
            host = ctx.global_view_selection(s, ref)
            monkeypatch.setenv("MI_DMRECON_GVS_DEVICE", "1")
            assert ctx.global_view_selection(s, ref) == host, (ref, gmax)


def test_c3_deterministic(c3, monkeypatch):
    cfg, scene, ctx, st, res, stats = c3
    refs = list(range(cfg["params"].n_views))
    # 20 views on the 8 XCDs: four XCDs hold three views (teams of 10), four hold two (teams of 16, given to the views with the
    # most empty pixels at the hand-over); with every team held to 10 workgroups the same bits
    assert stats["front_team"] == 10 and stats["front_team_max"] == 16, stats
    monkeypatch.setenv("MI_DMRECON_FRONT_TEAM", "10")
    uniform = ctx.reconstruct(st, refs, want_views=True)
    assert ctx.last_stats["front_team"] == 10 and ctx.last_stats["front_team_max"] == 10
    monkeypatch.delenv("MI_DMRECON_FRONT_TEAM")
    for v in range(len(refs)):
        for k in ("depth", "conf", "dz", "normal", "views"):
            assert np.array_equal(uniform[v][k], res[v][k]), (v, k)
    # the same call again, and the same call on a forked context (own stream): bit-identical maps
    f = ctx.fork()
    for c in (ctx, f):
        again = c.reconstruct(st, refs, want_views=True)
        for v in (0, 7, 19):
            for k in ("depth", "conf", "dz", "normal", "views"):
                assert np.array_equal(again[v][k], res[v][k]), (v, k)
    f.close()
    # A view's maps are the view's own: in another batch (three views instead of twenty) bit for bit the same -- the round
    # at which a view leaves the throughput lane layout is decided from the view's own list sizes
    sub = ctx.reconstruct(st, [0, 7, 19], want_views=True)
    for v, r in zip([0, 7, 19], sub):
        for k in ("depth", "conf", "dz", "normal", "views"):
            assert np.array_equal(r[k], res[v][k]), (v, k)
    assert stats["n_latency_rounds"] >= 1 and stats["n_bulk_launches"] > 10, stats      # both layouts ran in the big call


@pytest.mark.parametrize("view", [3, 8, 12])
def test_c3_views_vs_oracle(c3, c3_oracle, view):
    """Depth / confidence of three full-size views against the CPU oracle under the map-level tolerance.  The fill
    mask: IoU >= 0.98 as SURVEY 8c states -- except where the reference ALGORITHM does not reach that against itself:
    strips along the image border are filled or not depending on which local view set reaches them first, and the same
    restatement with its queue popped in another order (reversed, random, the reference's order with other tie-breaks:
    seven orders per view, tools/order_floor.py -> tests/golden/order_floor_c3.json, computed on the CPU) loses them
    too: against the reference's own order the worst of those orders reaches 0.9581 on view 12 and 0.9847 on view 8
    (0.9441 on view 14, 0.9696 on view 17, 0.9713 on view 13; 0.9938 on view 3).  The GPU sweep is one more re-ordering of
    the same algorithm: per view it must reach min(0.98, that view's floor - 0.002) -- measured: 0.9938 / 0.9856 / 0.9708 on
    views 3 / 8 / 12 -- and its depths 1.5 x the floor's relative depth p99 at most."""
    import json
    cfg, scene, ctx, st, res, stats = c3
    floors = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "order_floor_c3.json")))
    o = c3_oracle(view)
    m = map_parity(res[view]["depth"], res[view]["conf"], o["d"], o["c"])
    assert m["rel_med"] <= 1e-3 and m["rel_p99"] <= 5e-3, m
    assert m["conf_med"] <= 1e-3 and m["conf_p99"] <= 5e-3, m
    floor_iou = floors["worst"]["iou"][str(view)]
    assert m["iou"] >= min(0.98, floor_iou - 0.002), (m["iou"], floor_iou)
    assert m["rel_p99"] <= max(3e-3, 1.5 * floors["worst"]["rel_p99"][str(view)]), (m, floors["worst"]["rel_p99"][str(view)])
    if view == 12:
        # the fixture is what this restatement computes: its reversed order, recomputed here, against the stored figure
        r = c3_oracle(view, reverse=True)
        again = map_parity(r["d"], r["c"], o["d"], o["c"])
        assert abs(again["iou"] - floors["views"][str(view)]["orders"]["reverse"]["iou"]) < 1e-6, again


def test_c3_global_view_selection_of_view_3(c3):
    from oracle import oracle as orc
    cfg, scene, ctx, st, res, stats = c3
    assert ctx.global_view_selection(st, 3) == orc.OracleScene(scene).global_vs(orc.make_settings(ref_view=3, scale=cfg["scale"]))


@pytest.mark.parametrize("name,ref", [("C1", 0), ("C2", 5)])
def test_other_baseline_configs_vs_oracle(name, ref):
    """BASELINE configs 1 (2 views 640x480, scale 0, --local-neighbors=1) and 2 (8 views 1280x720, scale 1)
    at full size: one reference view against the CPU oracle, and the whole scene for sanity."""
    from oracle import oracle as orc
    cfg = CONFIGS[name]
    scene = make_scene(cfg["params"])
    ctx = api.Context(0)
    ctx.load_scene(scene)
    st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
    res = ctx.reconstruct(st, list(range(cfg["params"].n_views)))
    assert all((r["conf"] > 0).mean() > 0.5 for r in res)
    o = orc.OracleScene(scene).reconstruct(orc.make_settings(ref_view=ref, scale=cfg["scale"],
                                                              local_neighbors=cfg["local_neighbors"]))
    m = map_parity(res[ref]["depth"], res[ref]["conf"], o["depth"], o["conf"])
    assert m["iou"] >= 0.98 and m["rel_med"] <= 1e-3 and m["rel_p99"] <= 5e-3, m
    assert m["conf_med"] <= 1e-3 and m["conf_p99"] <= (1e-2 if cfg["local_neighbors"] == 1 else 5e-3), m
    ctx.close()


def test_config5_shape_large_images_async_staging():
    """BASELINE config 5 (4032x3024 images, scale 3) with a reduced view count: the pinned, asynchronous
    staging path must give the byte-identical pyramid of the blocking path, and a view reconstructed at
    scale 3 (504x378) must agree with the CPU oracle."""
    import time
    from oracle import oracle as orc
    from mve_amd.synth import SynthParams
    p = SynthParams(n_views=6, width=4032, height=3024, n_features=1500)
    scene = make_scene(p)
    a, b = api.Context(0), api.Context(0)
    t0 = time.time(); a.load_scene(scene); t_block = time.time() - t0
    t0 = time.time(); b.load_scene(scene, pinned_staging=True); t_async = time.time() - t0
    print("upload of 6 x 36.6 MB: blocking %.3f s, pinned/async %.3f s" % (t_block, t_async))
    assert a.num_levels(0) == 8 and a.level_size(0, 3) == (504, 378)
    for lvl in (0, 3, 7):
        assert np.array_equal(a.get_level(4, lvl)[0], b.get_level(4, lvl)[0])
    st = api.Settings(scale=3)
    res = b.reconstruct(st, list(range(6)))
    S = orc.OracleScene(scene)
    assert np.array_equal(b.get_level(2, 3)[0], S.pyramid_level(2, 3)[0])       # three Gaussian levels deep, byte-exact
    o = S.reconstruct(orc.make_settings(ref_view=2, scale=3))
    m = map_parity(res[2]["depth"], res[2]["conf"], o["depth"], o["conf"])
    assert m["iou"] >= 0.98 and m["rel_med"] <= 1e-3 and m["rel_p99"] <= 5e-3 and m["conf_p99"] <= 5e-3, m
    a.close(); b.close()


def test_c3_cancel_during_propagation(c3):
    """progress.cancelled raised for ONE view while the rounds run (dmrecon.cc:353): that view ends with
    RECON_CANCELLED and nothing of it is written (dmrecon.cc:101-105); the 19 views batched with it finish --
    also when the flag is first seen inside the blind tail."""
    import threading
    import time
    cfg, scene, ctx, st, res, stats = c3
    refs = list(range(cfg["params"].n_views))
    for wait_s in (0.0, 0.015):                      # cancel in the host-visible rounds / in the tail
        prog = (api.CProgress * len(refs))()
        out = ctx.alloc_outputs(st, refs, want_normal=False)
        for o in out:
            o["depth"].fill(-7.0)

        def canceller():
            t0 = time.time()
            while prog[0].status != 3 and time.time() - t0 < 10.0:      # MI_RECON_QUEUE: propagation running
                pass
            time.sleep(wait_s)
            prog[5].cancelled = 1

        th = threading.Thread(target=canceller)
        th.start()
        r = ctx.reconstruct(st, refs, want_normal=False, progress=prog, out=out)
        th.join()
        assert r[5]["status"] == api.E_CANCELLED and prog[5].status == 5     # RECON_CANCELLED
        assert (out[5]["depth"] == -7.0).all()                           # caller's buffers untouched
        for i in range(len(refs)):
            if i == 5:
                continue
            assert r[i]["status"] == 0 and prog[i].status == 0
            m = map_parity(r[i]["depth"], r[i]["conf"], res[i]["depth"], res[i]["conf"])
            assert m["iou"] >= 0.999 and m["rel_p99"] <= 3e-3, (i, m)
    # cancelling every view cancels the call
    prog = (api.CProgress * len(refs))()
    for p in prog:
        p.cancelled = 1
    with pytest.raises(InterruptedError):
        ctx.reconstruct(st, refs, want_normal=False, progress=prog)
    # the context is still usable and gives the same maps as before
    again = ctx.reconstruct(st, refs, want_views=True)
    assert np.array_equal(again[3]["depth"], res[3]["depth"])


# ---- BASELINE config 5 at its full size: 100 views of 4032 x 3024, scale 3 (504 x 378 maps) --------------------------

_ORACLE_C5 = """
import sys, numpy as np
sys.path.insert(0, %r)
from mve_amd.synth import CONFIGS, make_cameras, make_features, SceneData
from oracle import oracle as orc
cfg = CONFIGS["C5"]
p = cfg["params"]
cams = make_cameras(p)
imgs = np.load(sys.argv[2], mmap_mode="r")                      # the images the test process rendered
S = orc.OracleScene(SceneData(cams, [imgs[i] for i in range(p.n_views)], make_features(p, cams)))
out = {}
for v in sys.argv[3:]:                       # (one after the other: a reconstruction prepares the scene's views for itself)
    r = S.reconstruct(orc.make_settings(ref_view=int(v), scale=cfg["scale"], local_neighbors=cfg["local_neighbors"]))
    out["d" + v], out["c" + v] = r["depth"], r["conf"]
np.savez(sys.argv[1], **out)
"""


def test_config5_full_size_views_vs_oracle(tmp_path):
    """All 100 reference views of the config-5 scene in one call; views 0, 50 and 99 against the CPU oracle (one process
    next to the GPU work, on the images this process rendered: the three views one after the other).  The scene is rendered
    on the GPU (mve_amd/csrc/synth_render_gpu.hip, harness code: 100 x 12 MP take 87 s on the box's 16-CPU quota, well under a
    second there; GPU path and oracle read the same images).  Confidence bound 1e-2:
    the reference algorithm against itself with its queue reversed is at 6.0e-3 .. 7.8e-3 on this scene
    (tools/c5_order_floor.py, profiles/r4_c5_order_floor.json).  The three views alone give the bits they have in the
    batch of 100."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    views = (0, 50, 99)
    out = str(tmp_path / "c5_oracle.npz")
    proc = None
    try:
        import time
        cfg = CONFIGS["C5"]
        n = cfg["params"].n_views
        t0 = time.time(); scene = make_scene(cfg["params"], gpu=True); t1 = time.time()
        img_file = str(tmp_path / "c5_images.npy")
        np.save(img_file, np.stack(scene.images))
        proc = subprocess.Popen([sys.executable, "-c", _ORACLE_C5 % root, out, img_file] + [str(v) for v in views])
        ctx = api.Context(0)
        ctx.load_scene(scene, pinned_staging=True); t2 = time.time()
        st = api.Settings(scale=cfg["scale"], nrReconNeighbors=cfg["local_neighbors"])
        res = ctx.reconstruct(st, list(range(n))); t3 = time.time()
        print("C5: scene rendered in %.1f s, staged in %.1f s, 100 views reconstructed in %.2f s (first call)" % (t1 - t0, t2 - t1, t3 - t2))
        assert len(res) == n and res[0]["depth"].shape == (378, 504)
        for v in range(n):
            d, c = res[v]["depth"], res[v]["conf"]
            filled = c > 0
            assert np.isfinite(d).all() and (d[filled] > 0).all() and (d[~filled] == 0).all() and c.max() <= 1.0
            assert filled.mean() > 0.5, (v, filled.mean())
        again = ctx.reconstruct(st, list(views))
        for k, v in enumerate(views):
            for name in ("depth", "conf", "dz"):
                assert np.array_equal(again[k][name], res[v][name]), (v, name)
        assert proc.wait(timeout=1500) == 0
        print("C5: oracle process done %.1f s after the start" % (time.time() - t0))
        o = np.load(out)
        for v in views:
            m = map_parity(res[v]["depth"], res[v]["conf"], o["d%d" % v], o["c%d" % v])
            print("C5 view", v, m)
            assert m["iou"] >= 0.98 and m["rel_med"] <= 1e-3 and m["rel_p99"] <= 5e-3 and m["conf_p99"] <= 1e-2, (v, m)
            # regression guard (round 6): 1.25 x the worst of the three views as this build measures them (the maps are deterministic)
            assert m["iou"] >= 0.998 and m["rel_p99"] <= 2.9e-3 and m["conf_p99"] <= 8.9e-3, (v, m)
        ctx.close()
    finally:
        if proc is not None and proc.poll() is None:
            proc.kill()
