"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against
(a) fixtures produced by the real reference (tests/golden) and (b) the CPU oracle on the same
seeded inputs.

Stated tolerances (the noise floor of the reference against itself under a changed
queue order is median 3.5e-4 / p99 2.3e-3 relative depth, fill IoU 0.9999):

  patch level (same hypothesis, same view set)
      sampled colours abs <= 3e-5 (a sample position of ~150 px carries ~2 ulp = 3e-5 px of float
      rounding in either implementation, times a colour gradient of <1 per pixel),
      derivatives abs <= 1e-4 * max|deriv|, NCC abs <= 1e-4
      full doAutoOptimization: relative depth <= 1e-3 and |conf| <= 5e-3 on >= 99 % of patches,
      identical local view set on >= 98 %
  map level (parallel sweep vs sequential priority queue)
      fill-mask IoU >= 0.98; on pixels filled in both: relative depth median <= 1e-3, p99 <= 5e-3;
      confidence abs diff median <= 1e-3, p99 <= 5e-3
  pyramid: byte-exact; global view selection: identical
"""
import os
import numpy as np
import pytest

from conftest import map_parity
from mve_amd import api

pytestmark = pytest.mark.gpu

MAP_TOL = dict(iou=0.98, rel_med=1e-3, rel_p99=5e-3, conf_med=1e-3, conf_p99=5e-3)


def assert_map_parity(m, rel_p99=None):
    assert m["iou"] >= MAP_TOL["iou"], m
    assert m["rel_med"] <= MAP_TOL["rel_med"], m
    assert m["rel_p99"] <= (MAP_TOL["rel_p99"] if rel_p99 is None else rel_p99), m
    assert m["conf_med"] <= MAP_TOL["conf_med"], m
    assert m["conf_p99"] <= MAP_TOL["conf_p99"], m


@pytest.fixture(scope="module")
def ctx_g1(gpu_ctx, g1_scene):
    gpu_ctx.load_scene(g1_scene)
    return gpu_ctx


def test_pyramid_bytes_vs_oracle_and_reference(gpu_ctx, g1b, g1b_scene):
    from oracle import oracle as orc
    gpu_ctx.load_scene(g1b_scene)
    S = orc.OracleScene(g1b_scene)
    assert gpu_ctx.num_levels(2) == S.pyramid_levels(2)
    for lvl in range(gpu_ctx.num_levels(2)):
        img, proj, inv = gpu_ctx.get_level(2, lvl)
        oimg, oproj, oinv = S.pyramid_level(2, lvl)
        assert np.array_equal(img, oimg), "pyramid level %d differs" % lvl
        assert np.array_equal(proj, oproj) and np.array_equal(inv, oinv)
    # the reference's own "undist-L1" embedding
    assert np.array_equal(gpu_ctx.get_level(2, 1)[0], g1b["s1v2_undist"])


def test_pyramid_grey_alpha_and_odd_sizes(gpu_ctx):
    from mve_amd.scene_io import Camera, SceneData
    from oracle import oracle as orc
    rng = np.random.RandomState(3)
    cam = Camera(flen=0.9, paspect=1.0, ppoint=(0.5, 0.5), rot=[1, 0, 0, 0, 1, 0, 0, 0, 1], trans=[0, 0, 10])
    for (w, h, ch) in [(61, 47, 3), (128, 33, 1), (35, 90, 4), (31, 31, 2)]:
        img = rng.randint(0, 256, (h, w, ch)).astype(np.uint8)
        gpu_ctx.set_view(0, cam, img)
        rgb = img[:, :, :3] if ch >= 3 else np.repeat(img[:, :, :1], 3, axis=2)   # image_pyramid.cc:65-73
        S = orc.OracleScene(SceneData([cam], [np.ascontiguousarray(rgb)], []))
        assert gpu_ctx.num_levels(0) == S.pyramid_levels(0)
        for lvl in range(gpu_ctx.num_levels(0)):
            assert np.array_equal(gpu_ctx.get_level(0, lvl)[0], S.pyramid_level(0, lvl)[0]), (w, h, ch, lvl)


def test_global_view_selection(ctx_g1, g1):
    assert ctx_g1.global_view_selection(api.Settings(refViewNr=0)) == list(g1["gvs"])
    # with globalVSMax = 2 the greedy order decides
    from oracle import oracle as orc
    from conftest import scene_from_golden
    S = orc.OracleScene(scene_from_golden(g1))
    for ref in range(5):
        st = api.Settings(refViewNr=ref, globalVSMax=2)
        assert ctx_g1.global_view_selection(st) == S.global_vs(orc.make_settings(ref_view=ref, global_max=2))


def test_device_view_selection_equals_host_and_reference(gpu_ctx, g1, g1_scene, h1, h1_scene, monkeypatch):
    """gvs_device.hip (MI_DMRECON_GVS_DEVICE=1; the default for large calls, e.g. 100 views of a 100-view scene) selects exactly
    what the host loop and the reference select: every reference view of G1 and H1 (one low-overlap view, features
    outside frustums), tight globalVSMax where the greedy ORDER decides, and a bounding box that drops features."""
    from oracle import oracle as orc
    for scene, gold in ((g1_scene, g1), (h1_scene, h1)):
        gpu_ctx.load_scene(scene)
        S = orc.OracleScene(scene)
        nv = len(scene.cameras)
        box = dict(aabbMin=(-2.0, -1.5, -1e3), aabbMax=(1.0, 2.5, 1e3))
        for kw in (dict(), dict(globalVSMax=2), dict(globalVSMax=3, minParallax=25.0), box):
            for ref in range(nv):
                st = api.Settings(refViewNr=ref, **kw)
                monkeypatch.setenv("MI_DMRECON_GVS_DEVICE", "0")
                host = gpu_ctx.global_view_selection(st)
                monkeypatch.setenv("MI_DMRECON_GVS_DEVICE", "1")
                assert gpu_ctx.global_view_selection(st) == host, (kw, ref)
                so = orc.make_settings(ref_view=ref, global_max=kw.get("globalVSMax", 20), minParallax=kw.get("minParallax", 10.0))
                if "aabbMin" in kw:
                    so.aabbMin[:] = kw["aabbMin"]
                    so.aabbMax[:] = kw["aabbMax"]
                try:
                    want = S.global_vs(so)
                except Exception:
                    want = None                                       # the oracle reports "no view" as an error
                if want is not None:
                    assert host == want, (kw, ref)
        monkeypatch.setenv("MI_DMRECON_GVS_DEVICE", "1")
        assert gpu_ctx.global_view_selection(api.Settings(refViewNr=0)) == list(gold["gvs"])
        # a whole call planned on the device gives the maps of a call planned on the host
        refs = list(range(min(nv, 5)))
        dev = gpu_ctx.reconstruct(api.Settings(), refs, want_views=True)
        assert gpu_ctx.last_stats["gvs_on_device"] == 1
        monkeypatch.setenv("MI_DMRECON_GVS_DEVICE", "0")
        hst = gpu_ctx.reconstruct(api.Settings(), refs, want_views=True)
        assert gpu_ctx.last_stats["gvs_on_device"] == 0
        for a, b in zip(dev, hst):
            for k in ("depth", "conf", "dz", "views"):
                assert np.array_equal(a[k], b[k]), k
    # errors of a reference view stay per view
    monkeypatch.setenv("MI_DMRECON_GVS_DEVICE", "1")
    with pytest.raises(ValueError):
        gpu_ctx.global_view_selection(api.Settings(refViewNr=99))
    monkeypatch.delenv("MI_DMRECON_GVS_DEVICE")
    gpu_ctx.load_scene(g1_scene)                             # the scene the module's other tests expect (ctx_g1)


def test_patch_sampler_vs_reference_vectors(ctx_g1, g1):
    st = api.Settings(refViewNr=0)
    n_checked = 0
    for i in range(len(g1["seeds_xy"])):
        x, y = [int(v) for v in g1["seeds_xy"][i]]
        d, dzi, dzj = [float(v) for v in g1["seeds_hyp"][i]]
        e = ctx_g1.patch_eval(st, 0, x, y, d, dzi, dzj)
        assert e["master"][0] == g1["ev_master"][i, 0]
        if not e["master"][0]:
            continue
        assert abs(e["master"][1] - g1["ev_master"][i, 1]) <= 1e-6          # masterMeanCol
        assert np.abs(e["master"][2:] - g1["ev_master"][i, 2:]).max() <= 1e-4   # patch normal
        assert np.array_equal(e["ok"], g1["ev_ok"][i])
        okv = e["ok"] > 0
        if not okv.any():
            continue
        assert np.abs(e["ncc"][okv] - g1["ev_ncc"][i][okv]).max() <= 1e-4
        assert np.abs(e["col"][okv] - g1["ev_col"][i][okv]).max() <= 3e-5
        dref = g1["ev_der"][i][okv]
        assert np.abs(e["deriv"][okv] - dref).max() <= 1e-4 * max(np.abs(dref).max(), 1.0)
        n_checked += int(okv.sum())
    assert n_checked > 50


def test_patch_optimization_vs_reference_vectors(ctx_g1, g1):
    out, loc = ctx_g1.patch_optimize(api.Settings(refViewNr=0), 0, g1["seeds_xy"], g1["seeds_hyp"], g1["seeds_local"])
    ref, ref_loc = g1["opt"], g1["opt_local"]
    assert (out[:4, 0] == 0).all()                                           # border patches fail
    agree = (out[:, 0] > 0) == (ref[:, 0] > 0)
    assert agree.mean() >= 0.98
    ok = (out[:, 0] > 0) & (ref[:, 0] > 0)
    assert ok.sum() >= 10
    rel = np.abs(out[ok, 1] - ref[ok, 1]) / ref[ok, 1]
    assert (rel <= 1e-3).mean() >= 0.99
    assert (np.abs(out[ok, 0] - ref[ok, 0]) <= 5e-3).mean() >= 0.99
    assert (loc[ok] == ref_loc[ok]).all(1).mean() >= 0.98


def test_patch_optimization_vs_oracle_many(ctx_g1, g1_scene):
    from oracle import oracle as orc
    S = orc.OracleScene(g1_scene)
    rng = np.random.RandomState(11)
    n = 600
    xy = np.stack([rng.randint(2, 158, n), rng.randint(2, 118, n)], 1)
    hyp = np.stack([10.0 + rng.uniform(-0.4, 0.4, n), rng.uniform(-5e-3, 5e-3, n), rng.uniform(-5e-3, 5e-3, n)], 1)
    for ref in (0, 3):
        go, gl = ctx_g1.patch_optimize(api.Settings(refViewNr=ref), ref, xy, hyp)
        oo, ol = S.patch_optimize(orc.make_settings(ref_view=ref), xy, hyp)
        assert ((go[:, 0] > 0) == (oo[:, 0] > 0)).mean() >= 0.98
        ok = (go[:, 0] > 0) & (oo[:, 0] > 0)
        assert ok.sum() > 50
        rel = np.abs(go[ok, 1] - oo[ok, 1]) / oo[ok, 1]
        assert (rel <= 1e-3).mean() >= 0.99
        assert (np.abs(go[ok, 0] - oo[ok, 0]) <= 5e-3).mean() >= 0.99
        assert (gl[ok] == ol[ok]).all(1).mean() >= 0.98
        assert (np.abs(go[ok, 4:7] - oo[ok, 4:7]).max(1) <= 1e-3).mean() >= 0.98   # normals


def test_lane_layouts_agree(ctx_g1, g1):
    # the tail rounds run one patch per wavefront (16 lanes per view): same maths as the 16-patch throughput layout,
    # different lane layout and summation order
    st = api.Settings(refViewNr=0)
    a, al = ctx_g1.patch_optimize(st, 0, g1["seeds_xy"], g1["seeds_hyp"], g1["seeds_local"], lanes_per_view=1)
    b, bl = ctx_g1.patch_optimize(st, 0, g1["seeds_xy"], g1["seeds_hyp"], g1["seeds_local"], lanes_per_view=16)
    assert np.array_equal(a[:, 0] > 0, b[:, 0] > 0)
    ok = a[:, 0] > 0
    assert np.abs(a[ok, 1] - b[ok, 1]).max() / 10.0 <= 1e-5      # only the summation order differs
    assert np.abs(a[ok, 0] - b[ok, 0]).max() <= 1e-4
    assert np.array_equal(al[ok], bl[ok]) and np.array_equal(a[ok, 7], b[ok, 7])
    ref = g1["opt"]
    okr = (b[:, 0] > 0) & (ref[:, 0] > 0)
    assert (np.abs(b[okr, 1] - ref[okr, 1]) / ref[okr, 1] <= 1e-3).mean() >= 0.99


def test_speculative_tail_equals_sequential_attempts(gpu_ctx, g1_scene, monkeypatch):
    """The tail rounds run a pixel's candidate hypotheses in parallel and apply the reference's sequential rule
    (pop-time skip dmrecon.cc:371, accept-if-better :391) afterwards.  Host-visible rounds in the same lane layout
    run them one after the other: the maps must be bit-identical."""
    gpu_ctx.load_scene(g1_scene)
    st = api.Settings()
    monkeypatch.setenv("MI_DMRECON_FRONT", "0")                        # (the front kernel has its own test)
    monkeypatch.setenv("MI_DMRECON_VIEW_HANDOVER", "1000000000")       # every view in the latency layout from the first round on
    monkeypatch.setenv("MI_DMRECON_HOST_ROUNDS", "1")                  # never enter the tail
    seq = gpu_ctx.reconstruct(st, [0, 1, 2, 3, 4], want_views=True)
    n_seq = dict(gpu_ctx.last_stats)
    monkeypatch.delenv("MI_DMRECON_HOST_ROUNDS")                       # tail rounds from the first round on
    spec = gpu_ctx.reconstruct(st, [0, 1, 2, 3, 4], want_views=True)
    n_spec = dict(gpu_ctx.last_stats)
    assert n_spec["n_tail_launches"] >= 10 and n_seq["n_tail_launches"] == 0
    for a, b in zip(seq, spec):
        for k in ("depth", "conf", "dz", "normal", "views"):
            assert np.array_equal(a[k], b[k]), k
    # the device counts the attempts the reference's rule would have made, not the speculative extras
    assert n_seq["n_patch"] == n_spec["n_patch"] and n_seq["n_eval"] == n_spec["n_eval"]


def test_sparse_maps_equal_full_copies(gpu_ctx, g1_scene, h1_scene, monkeypatch):
    """Large batches get their maps as a snapshot taken at the hand-over to the front kernel (copied while the kernel
    runs) plus the list of the pixels the front changed afterwards, written over the snapshot by the host
    (BatchRun::front_rounds, k_emit_changed).  The maps are the ones a full copy after the kernel gives, bit for bit --
    with and without the normal map (records of 9 / 6 words), for views listed several times (a merged batch's callers
    may ask for the same view), and when the list outgrows its buffer (everything is copied in full then)."""
    st = api.Settings()
    # (views of 160 x 120 / 208 x 156 pixels: at the default hand-over -- a view's own list below 320 entries -- the front kernel
    # would do most of their propagation and its list of changed pixels would be most of the image; handed over later, the
    # list is the small part it is on full-size views: C3 3.7 % of the pixels)
    monkeypatch.setenv("MI_DMRECON_VIEW_HANDOVER", "40")
    for scene, refs in ((g1_scene, [0, 1, 2, 3, 4, 2, 0]), (h1_scene, list(range(9)))):
        gpu_ctx.load_scene(scene)
        monkeypatch.setenv("MI_DMRECON_SPARSE_MAPS", "0")
        full = gpu_ctx.reconstruct(st, refs, want_normal=True)
        assert gpu_ctx.last_stats["n_sparse_records"] == 0
        for want_normal in (True, False):
            monkeypatch.setenv("MI_DMRECON_SPARSE_MAPS", "1")
            got = gpu_ctx.reconstruct(st, refs, want_normal=want_normal)
            s = dict(gpu_ctx.last_stats)
            assert s["n_front_launches"] == 1 and s["n_sparse_records"] > 5, s        # the path ran
            assert s["n_sparse_records"] < 0.6 * s["n_filled"], s                       # ... and the list is the smaller part
            for a, b in zip(got, full):
                for k in ("depth", "conf", "dz") + (("normal",) if want_normal else ()):
                    assert np.array_equal(a[k], b[k]), (k, want_normal)
        # a list that does not fit (test hook: room for 5 records): full copies after all
        monkeypatch.setenv("MI_DMRECON_SPARSE_MAPS", "1")
        monkeypatch.setenv("MI_DMRECON_DEBUG_SPARSE_CAP", "5")
        got = gpu_ctx.reconstruct(st, refs, want_normal=True)
        assert gpu_ctx.last_stats["n_sparse_records"] == -1
        monkeypatch.delenv("MI_DMRECON_DEBUG_SPARSE_CAP")
        for a, b in zip(got, full):
            for k in ("depth", "conf", "dz", "normal"):
                assert np.array_equal(a[k], b[k]), (k, "overflow")
        monkeypatch.delenv("MI_DMRECON_SPARSE_MAPS")
    monkeypatch.delenv("MI_DMRECON_VIEW_HANDOVER")


def test_a_view_that_fails_in_the_front_phase_on_the_sparse_path(gpu_ctx, g1_scene, monkeypatch):
    """mi_dmrecon.h: the maps of a view that does not finish stay untouched -- except in a large batch (the snapshot + changed
    pixels path), where a view that ends with an error AFTER the hand-over has had the snapshot written: its status says the
    maps are void, and the other views of the batch are what they are without the failure.  Test hook
    MI_DMRECON_DEBUG_FRONT_EFOOTPRINT: the view's footprint flag is raised behind the snapshot copies."""
    st = api.Settings()
    gpu_ctx.load_scene(g1_scene)
    monkeypatch.setenv("MI_DMRECON_VIEW_HANDOVER", "40")
    monkeypatch.setenv("MI_DMRECON_SPARSE_MAPS", "1")
    refs = [0, 1, 2, 3, 4]
    clean = gpu_ctx.reconstruct(st, refs, want_normal=False)
    assert gpu_ctx.last_stats["n_sparse_records"] > 5 and all(m["status"] == 0 for m in clean)
    monkeypatch.setenv("MI_DMRECON_DEBUG_FRONT_EFOOTPRINT", "2")
    out = gpu_ctx.alloc_outputs(st, refs, want_normal=False)
    for m in out:
        m["depth"][:] = -7.0                                   # a canary: what "untouched" would look like
    got = gpu_ctx.reconstruct(st, refs, want_normal=False, out=out)
    monkeypatch.delenv("MI_DMRECON_DEBUG_FRONT_EFOOTPRINT")
    assert [m["status"] for m in got] == [0, 0, api.E_FOOTPRINT, 0, 0]
    for i in (0, 1, 3, 4):
        for k in ("depth", "conf", "dz"):
            assert np.array_equal(got[i][k], clean[i][k]), (i, k)
    # the failed view: documented as void; what it holds is the state at the hand-over (no canary left, no pixel of a later round
    # is required) -- with the strict form (MI_DMRECON_SPARSE_MAPS=0) it is untouched
    assert not np.any(got[2]["depth"] == -7.0)
    monkeypatch.setenv("MI_DMRECON_SPARSE_MAPS", "0")
    monkeypatch.setenv("MI_DMRECON_DEBUG_FRONT_EFOOTPRINT", "2")       # (no snapshot: the hook has nothing to stand behind and is not reached)
    for m in out:
        m["depth"][:] = -7.0
    got = gpu_ctx.reconstruct(st, refs, want_normal=False, out=out)
    assert all(m["status"] == 0 for m in got)
    monkeypatch.delenv("MI_DMRECON_DEBUG_FRONT_EFOOTPRINT")
    monkeypatch.delenv("MI_DMRECON_SPARSE_MAPS")
    monkeypatch.delenv("MI_DMRECON_VIEW_HANDOVER")


@pytest.mark.parametrize("per_view,team", [("all", "1"), ("1000000", "1"), ("2", "1"), (None, None),
                                           ("all", "8"), ("all", "3"), ("1000000", "25"), ("2", "2")])
def test_front_kernel_equals_one_launch_per_round(gpu_ctx, g1_scene, h1_scene, monkeypatch, per_view, team):
    """k_front (the end of the tail: one persistent workgroup per reference view, each view at its own pace) writes
    exactly what one k_tail launch per round writes: same candidates, same attempts, same sequential rule
    (dmrecon.cc:365-431).  "all": the whole propagation after the first round in k_front (rounds with hundreds of
    entries per view, chunked); then handed over after the bulk rounds, late (two entries per view), and at the default
    threshold.  Five / nine reference views per call; on the hard scene views fail patches, replace local views and
    end at very different rounds.  team: workgroups per view (the attempts of a round dealt out over them, results
    exchanged through the view's mailbox; default: eight for a call that has the GPU to itself)."""
    if team is not None:
        monkeypatch.setenv("MI_DMRECON_FRONT_TEAM", team)
    for scene, refs in ((g1_scene, [0, 1, 2, 3, 4]), (h1_scene, list(range(9)))):
        gpu_ctx.load_scene(scene)
        monkeypatch.setenv("MI_DMRECON_FRONT", "0")
        if per_view == "all":
            monkeypatch.setenv("MI_DMRECON_VIEW_HANDOVER", "1000000000")       # tail rounds from the first round on
        ref = gpu_ctx.reconstruct(api.Settings(), refs, want_views=True)
        s0 = dict(gpu_ctx.last_stats)
        assert s0["n_front_launches"] == 0
        if per_view is None:
            monkeypatch.delenv("MI_DMRECON_FRONT")
        else:
            monkeypatch.setenv("MI_DMRECON_FRONT", "1000000" if per_view == "all" else per_view)
        for rep in range(2):
            got = gpu_ctx.reconstruct(api.Settings(), refs, want_views=True)
            s1 = dict(gpu_ctx.last_stats)
            if per_view == "all":
                # (the host-visible rounds are enqueued one ahead of their read-back: the hand-over comes a round late)
                assert s1["n_front_launches"] == 1 and s1["front_first_round"] in (3, 4) and s1["n_tail_launches"] == 0, s1   # (the propagation starts with round 2: seed_round)
                assert s1["n_front_rounds_max"] > 5 and s1["n_front_views"] == len(refs), s1
            if s1["n_front_launches"]:
                # default: the CUs of an XCD (32) dealt over the views that share it (views are dealt over the 8 XCDs)
                assert s1["front_team"] == (min(32, 32 // ((len(refs) + 7) // 8)) if team is None else min(int(team), 32 // ((len(refs) + 7) // 8))), s1
            assert s1["n_rounds"] == s0["n_rounds"], (s1["n_rounds"], s0["n_rounds"], s1["front_first_round"])
            for k in ("n_patch", "n_eval", "n_filled"):
                assert s1[k] == s0[k], k
            for a, b in zip(got, ref):
                for k in ("depth", "conf", "dz", "normal", "views"):
                    assert np.array_equal(a[k], b[k]), (k, rep)
        if team == "1":
            # one workgroup per view: the order in which the views' workgroups start (default: the most empty pixels first)
            # is scheduling only
            monkeypatch.setenv("MI_DMRECON_FRONT_ORDER", "0")
            got = gpu_ctx.reconstruct(api.Settings(), refs, want_views=True)
            monkeypatch.delenv("MI_DMRECON_FRONT_ORDER")
            for a, b in zip(got, ref):
                for k in ("depth", "conf", "dz", "normal", "views"):
                    assert np.array_equal(a[k], b[k]), (k, "front order")
        monkeypatch.delenv("MI_DMRECON_VIEW_HANDOVER", raising=False)
    monkeypatch.delenv("MI_DMRECON_FRONT", raising=False)
    monkeypatch.delenv("MI_DMRECON_FRONT_TEAM", raising=False)
    gpu_ctx.load_scene(g1_scene)                             # the scene the module's other tests expect (ctx_g1)


@pytest.mark.parametrize("fault", ["3", "0:7", "5:3"])
def test_front_team_gives_up_and_the_views_finish(gpu_ctx, g1_scene, h1_scene, monkeypatch, fault):
    """A front team needs all its workgroups on the GPU at once; what else holds compute units (another process, another
    program) the library cannot see.  A member that is not there in time must not fail the call: the team gives up, and
    the views go on from the round they stopped in with one workgroup each -- same maps as without teams, bit for bit.
    The test hook makes one member of every team vanish: never there ("3"), or after 7 / 3 rounds of its view."""
    monkeypatch.setenv("MI_DMRECON_VIEW_HANDOVER", "1000000000")           # the whole propagation in the front kernel
    monkeypatch.setenv("MI_DMRECON_FRONT", "1000000")
    monkeypatch.setenv("MI_DMRECON_TEAM_WAIT_US", "3000")
    for scene, refs in ((g1_scene, [0, 1, 2, 3, 4]), (h1_scene, list(range(9)))):
        gpu_ctx.load_scene(scene)
        monkeypatch.setenv("MI_DMRECON_FRONT_TEAM", "1")
        ref = gpu_ctx.reconstruct(api.Settings(), refs, want_views=True)
        s0 = dict(gpu_ctx.last_stats)
        assert s0["front_fallbacks"] == 0 and s0["front_team"] == 1
        monkeypatch.setenv("MI_DMRECON_FRONT_TEAM", "8")
        monkeypatch.setenv("MI_DMRECON_DEBUG_FRONT_FAULT", fault)
        got = gpu_ctx.reconstruct(api.Settings(), refs, want_views=True)
        s1 = dict(gpu_ctx.last_stats)
        monkeypatch.delenv("MI_DMRECON_DEBUG_FRONT_FAULT")
        assert s1["front_fallbacks"] == 1 and s1["front_team"] == 8, s1
        assert s1["n_filled"] == s0["n_filled"], (s1["n_filled"], s0["n_filled"])
        for a, b in zip(got, ref):
            for k in ("depth", "conf", "dz", "normal", "views"):
                assert np.array_equal(a[k], b[k]), (k, fault)
        # ... and the next call on the same context (same mailboxes, same flags) runs its teams undisturbed -- here as a
        # team does whose workgroups were NOT all placed on one XCD (it writes its state through the L2s)
        monkeypatch.setenv("MI_DMRECON_DEBUG_TEAM_WT", "1")
        again = gpu_ctx.reconstruct(api.Settings(), refs, want_views=True)
        monkeypatch.delenv("MI_DMRECON_DEBUG_TEAM_WT")
        assert gpu_ctx.last_stats["front_fallbacks"] == 0 and gpu_ctx.last_stats["front_team"] == 8
        for a, b in zip(again, ref):
            assert np.array_equal(a["depth"], b["depth"]) and np.array_equal(a["conf"], b["conf"])
    for k in ("MI_DMRECON_VIEW_HANDOVER", "MI_DMRECON_FRONT", "MI_DMRECON_TEAM_WAIT_US", "MI_DMRECON_FRONT_TEAM"):
        monkeypatch.delenv(k)
    gpu_ctx.load_scene(g1_scene)


def test_team_token_is_exclusive_per_gpu(gpu_ctx, g1_scene, monkeypatch):
    """Only one call per GPU runs front teams at a time, across processes: an advisory lock file named after the GPU's
    PCI address.  Held by somebody else (here: this test, through a second open of the file), a call runs without teams
    -- and gets the same maps."""
    import fcntl
    import glob
    gpu_ctx.load_scene(g1_scene)
    monkeypatch.setenv("MI_DMRECON_VIEW_HANDOVER", "1000000000")
    monkeypatch.setenv("MI_DMRECON_FRONT", "1000000")
    ref = gpu_ctx.reconstruct(api.Settings(), [0, 1, 2, 3, 4])
    assert gpu_ctx.last_stats["front_team"] > 1
    locks = glob.glob("/dev/shm/mi_dmrecon_team_*.lock") + glob.glob("/tmp/mi_dmrecon_team_*.lock")
    assert locks, "the team launch above must have created its lock file"
    held = [open(p, "r") for p in locks]                # (read-only is enough for flock, and all the library itself asks for)
    for f in held:
        fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)        # free again after the call: the lock is per front launch
    got = gpu_ctx.reconstruct(api.Settings(), [0, 1, 2, 3, 4])
    assert gpu_ctx.last_stats["front_team"] == 1 and gpu_ctx.last_stats["front_fallbacks"] == 0
    for f in held:
        f.close()
    for a, b in zip(got, ref):
        assert np.array_equal(a["depth"], b["depth"]) and np.array_equal(a["conf"], b["conf"])
    monkeypatch.delenv("MI_DMRECON_VIEW_HANDOVER"); monkeypatch.delenv("MI_DMRECON_FRONT")


def test_maps_do_not_depend_on_the_batch(gpu_ctx, g1_scene, h1_scene, monkeypatch):
    """A reference view's maps are the view's own: alone in a call, with four others, in a call merged with other calls,
    with one launch per throughput round or two -- bit for bit the same (as the reference's all-views mode writes what
    `-m ID` writes, apps/dmrecon/dmrecon.cc:285-318).  What could change the last bits -- the round at which the view's
    patches move from the throughput to the latency lane layout, which sum in different orders -- is decided per view
    from the view's own list sizes (k_generate), with a threshold low enough here that views hand over at different
    rounds of one batch."""
    import threading
    st = api.Settings()
    for scene, refs, handover in ((g1_scene, [0, 1, 2, 3, 4], "150"), (h1_scene, list(range(9)), "60"), (g1_scene, [0, 1, 2, 3, 4], None)):
        gpu_ctx.load_scene(scene)
        if handover:
            monkeypatch.setenv("MI_DMRECON_VIEW_HANDOVER", handover)
        else:
            monkeypatch.delenv("MI_DMRECON_VIEW_HANDOVER", raising=False)
        ref = gpu_ctx.reconstruct(st, refs, want_views=True)
        s_all = dict(gpu_ctx.last_stats)
        if handover:
            assert s_all["n_latency_rounds"] >= 1 and s_all["n_bulk_launches"] >= s_all["n_latency_rounds"] + 3, s_all   # both layouts ran
        # one view per call; sub-batches; another order
        for v in refs[:3]:
            one = gpu_ctx.reconstruct(st, [v], want_views=True)[0]
            for k in ("depth", "conf", "dz", "normal", "views"):
                assert np.array_equal(one[k], ref[refs.index(v)][k]), (v, k)
        sub = gpu_ctx.reconstruct(st, [refs[3], refs[1]], want_views=True)
        for r, v in zip(sub, (refs[3], refs[1])):
            for k in ("depth", "conf", "dz", "normal", "views"):
                assert np.array_equal(r[k], ref[refs.index(v)][k]), (v, k)
        # the throughput rounds as first-attempt launch + follow-up launch instead of one launch, without the speculative
        # small rounds, and with every round speculative (records for 2 x the threshold; larger rounds fall back on the
        # device): same arithmetic, same counters
        # (ONE_LAUNCH=0: first attempts, then one launch per further attempt; with SINGLE_FOLLOW=0 one follow-up launch that
        # runs an entry's remaining attempts in a row)
        for env in ({"MI_DMRECON_ONE_LAUNCH": "0", "MI_DMRECON_SPEC_ROUNDS": "0"}, {"MI_DMRECON_SPEC_ROUNDS": "0"},
                    {"MI_DMRECON_ONE_LAUNCH": "0", "MI_DMRECON_SPEC_ROUNDS": "0", "MI_DMRECON_SINGLE_FOLLOW": "0"},
                    {"MI_DMRECON_ONE_LAUNCH": "0", "MI_DMRECON_SPEC_ROUNDS": "0", "MI_DMRECON_SINGLE_FOLLOW": "1"},
                    {"MI_DMRECON_SPEC_ROUNDS": "1000000"}, {"MI_DMRECON_SPEC_ROUNDS": "700", "MI_DMRECON_ONE_LAUNCH": "0"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            two = gpu_ctx.reconstruct(st, refs, want_views=True)
            s_two = dict(gpu_ctx.last_stats)
            for k in env:
                monkeypatch.delenv(k)
            for a, b in zip(two, ref):
                for k in ("depth", "conf", "dz", "normal", "views"):
                    assert np.array_equal(a[k], b[k]), (k, env)
            # (the views replaced are counted for the attempts the reference's rule makes: a speculative attempt the rule
            # discards only NOTES what it would count -- or the footprint exception that would end its view -- and
            # k_apply_spec carries that out for the attempts that are consumed)
            for k in ("n_patch", "n_eval", "n_filled", "n_rounds", "n_view_replaced", "n_iter14"):
                assert s_two[k] == s_all[k], (k, env, s_two[k], s_all[k])
        # calls that meet inside the library are merged into one batch (default): every caller gets its own call's maps
        forks = [gpu_ctx.fork() for _ in range(3)]
        parts = [refs[:2], refs[2:3], refs[3:]]
        out = [None] * 3
        go = threading.Barrier(3)

        def worker(i):
            go.wait()
            out[i] = forks[i].reconstruct(st, parts[i], want_views=True)

        for _ in range(2):
            th = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for part, res in zip(parts, out):
                for v, r in zip(part, res):
                    for k in ("depth", "conf", "dz", "normal", "views"):
                        assert np.array_equal(r[k], ref[refs.index(v)][k]), (v, k)
        for f in forks:
            f.close()
    monkeypatch.delenv("MI_DMRECON_VIEW_HANDOVER", raising=False)
    gpu_ctx.load_scene(g1_scene)


def test_more_seeds_than_pixels(gpu_ctx):
    """Coarse scales of small images: a call's seed list can be longer than its pixels (many SfM features per reference
    view, few pixels).  The list buffers are sized for it in the same step as the state maps -- growing them afterwards
    used to reallocate the maps the jobs already pointed at."""
    from mve_amd.synth import SynthParams, make_scene
    sc = make_scene(SynthParams(n_views=5, width=160, height=120, n_features=6000))
    c = api.Context(0)                                          # a fresh context: fresh (unsized) scratch sets
    c.load_scene(sc)
    st = api.Settings(scale=2)                                  # 40 x 30 = 1 200 pixels per view, ~6 000 seeds each
    a = c.reconstruct(st, [0, 1, 2], want_views=True)
    assert c.last_stats["n_seeds"] > 2 * 3 * 40 * 30 + 64, c.last_stats["n_seeds"]
    b = c.reconstruct(st, [0, 1, 2], want_views=True)           # second call: buffers already large enough
    for x, y in zip(a, b):
        for k in ("depth", "conf", "dz", "views"):
            assert np.array_equal(x[k], y[k]), k
    assert all((x["conf"] > 0).mean() > 0.2 for x in a)
    c.close()


def test_batch_scratch_is_leased_from_the_scene(gpu_ctx, g1_scene, monkeypatch):
    """Lists, results and state maps of a batch belong to the scene, not to the context that runs the batch: calls that
    follow each other on different forked contexts re-use ONE set (whichever context leads a merged batch inside a timed
    region finds the buffers of the largest batch so far), calls that overlap get one each; the maps do not depend on
    which set a call got."""
    import threading
    gpu_ctx.load_scene(g1_scene)
    monkeypatch.setenv("MI_DMRECON_MERGE_CALLS", "0")
    st = api.Settings()
    ref = gpu_ctx.reconstruct(st, [0, 1, 2, 3, 4], want_views=True)
    sets0, px0 = gpu_ctx.debug_scratch_sets()
    assert sets0 >= 1 and px0 >= 5 * 160 * 120
    forks = [gpu_ctx.fork() for _ in range(3)]
    for f in forks:                                         # one after the other: no new set, none grows
        got = f.reconstruct(st, [2], want_views=True)
        assert np.array_equal(got[0]["depth"], gpu_ctx.reconstruct(st, [2], want_views=True)[0]["depth"])
        assert f.debug_scratch_sets() == (sets0, px0)
    go = threading.Barrier(3)
    out = [None] * 3

    def worker(i):
        go.wait()
        out[i] = forks[i].reconstruct(st, [0, 1, 2, 3, 4], want_views=True)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sets1, px1 = gpu_ctx.debug_scratch_sets()
    assert sets0 <= sets1 <= max(sets0, 3) and px1 == px0   # at most one per overlapping call, all back in the pool
    for o in out:
        for a, b in zip(o, ref):
            for k in ("depth", "conf", "dz", "normal", "views"):
                assert np.array_equal(a[k], b[k]), k
    for f in forks:
        f.close()


def test_concurrent_calls_are_merged_and_keep_their_own_results(gpu_ctx, g1_scene, monkeypatch):
    """Calls that arrive together on forked contexts of one scene with equal settings run as ONE batch
    (mi_dmrecon_reconstruct's merge front end): every caller gets the maps, statuses and errors of its own call; the
    statistics sit with the call that ran the batch; a call with other settings is not mixed in."""
    import threading
    gpu_ctx.load_scene(g1_scene)
    st = api.Settings()
    st3 = api.Settings(globalVSMax=3)
    alone = {v: gpu_ctx.reconstruct(st, [v], want_views=True)[0] for v in range(5)}
    alone3 = gpu_ctx.reconstruct(st3, [1, 2], want_views=True)
    assert gpu_ctx.last_stats["n_merged_calls"] == 1 and gpu_ctx.last_stats["merged_into_other_call"] == 0
    monkeypatch.setenv("MI_DMRECON_MERGE_WINDOW_US", "50000")           # the leader waits 50 ms: everybody joins
    # view 4 gets a non-positive pixel footprint (fault injection): the call that asks for nothing else fails with the
    # reference's exception, the call that asks for 4 among others gets its other views
    api.debug_inject_footprint(4)
    plans = [(st, [0, 1]), (st, [2]), (st, [3, 1, 0]), (st, [4]), (st3, [1, 2]), (st, [3, 4])]
    forks = [gpu_ctx.fork() for _ in plans]
    go = threading.Barrier(len(plans))
    out = [None] * len(plans)

    def worker(i):
        go.wait()
        try:
            out[i] = ("ok", forks[i].reconstruct(plans[i][0], plans[i][1], want_views=True), dict(forks[i].last_stats))
        except Exception as e:                                          # noqa: BLE001 - the exception IS the result
            out[i] = ("err", e, None)

    for rep in range(2):
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(plans))]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert out[3][0] == "err" and isinstance(out[3][1], IndexError), out[3]    # std::out_of_range, its only view
        for i in (0, 1, 2, 5):
            assert out[i][0] == "ok", out[i]
            for v, r in zip(plans[i][1], out[i][1]):
                if v == 4:
                    assert r["status"] == api.E_FOOTPRINT
                    continue
                assert r["status"] == 0
                for k in ("depth", "conf", "dz", "normal", "views"):
                    assert np.array_equal(r[k], alone[v][k]), (i, v, k)
        assert out[4][0] == "ok"
        for a, b in zip(out[4][1], alone3):
            for k in ("depth", "conf", "views"):
                assert np.array_equal(a[k], b[k]), k
        served = sorted(o[2]["n_merged_calls"] for o in out if o[0] == "ok")
        followers = sum(o[2]["merged_into_other_call"] for o in out if o[0] == "ok")
        # five calls with equal settings: once calls of the scene have met (second repetition at the latest) the first
        # to arrive waits for the others and all run in one batch -- four followers, of which the failing call shows
        # no statistics.  In the first repetition nobody expects company yet: the first TWO arrivals may each have started
        # alone (two batches of a scene run side by side), the third leads the rest -- two followers, one of which can be
        # the failing call: at least one that shows
        # ... and in the second one the leader stops gathering as soon as as many calls are there as the batches before it saw
        # (MergeQueue::expect): a batch of at least three of the five -- led by a call that shows its statistics, or by the failing
        # call, whose two or more followers then do
        assert max(served) <= 5 and followers >= 1 and (rep == 0 or max(served) >= 3 or followers >= 2), (rep, served, followers)
        assert out[4][2]["n_merged_calls"] <= 1                                   # other settings: its own batch
    # ... and deterministically: the leader gathers until all six calls are there (test hook; the window is 50 ms) -- the five calls
    # with equal settings are ONE batch: the leader shows five served calls, or -- if the failing call leads, which shows no
    # statistics -- all four others are followers
    monkeypatch.setenv("MI_DMRECON_DEBUG_MERGE_WAIT_FOR", str(len(plans)))
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(plans))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    monkeypatch.delenv("MI_DMRECON_DEBUG_MERGE_WAIT_FOR")
    ok5 = [o for i, o in enumerate(out) if i != 4 and o[0] == "ok"]
    served = sorted(o[2]["n_merged_calls"] for o in ok5)
    followers = sum(o[2]["merged_into_other_call"] for o in ok5)
    if out[4][2]["n_merged_calls"] == 1:          # (unless the call with other settings happened to lead the gathering: then the five
        assert (served[-1] == 5 and followers == 3) or (served[-1] == 0 and followers == 4), (served, followers)   # form by themselves)
    for i in (0, 1, 2, 5):
        for v, r in zip(plans[i][1], out[i][1]):
            if v != 4:
                for k in ("depth", "conf", "dz"):
                    assert np.array_equal(r[k], alone[v][k]), (i, v, k)
    monkeypatch.setenv("MI_DMRECON_MERGE_CALLS", "0")
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(plans))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert all(o[2]["merged_into_other_call"] == 0 for o in out if o[0] == "ok")
    api.debug_inject_footprint(-1)
    for f in forks:
        f.close()


def test_views_end_individually_in_a_batch(gpu_ctx, g1_scene, monkeypatch):
    """A cancelled view or a view whose footprint turns non-positive (patch_sampler.cc:78-82 throws) ends alone;
    the other views of the call finish, with the maps they get without it (apps/dmrecon/dmrecon.cc:314-317)."""
    gpu_ctx.load_scene(g1_scene)
    st = api.Settings()
    base = gpu_ctx.reconstruct(st, [0, 1, 2, 3, 4])
    # (a) view 2 cancelled before the start
    prog = (api.CProgress * 5)()
    prog[2].cancelled = 1
    out = gpu_ctx.alloc_outputs(st, [0, 1, 2, 3, 4])
    out[2]["depth"].fill(-7.0)
    res = gpu_ctx.reconstruct(st, [0, 1, 2, 3, 4], progress=prog, out=out)
    assert [r["status"] for r in res] == [0, 0, api.E_CANCELLED, 0, 0]
    assert prog[2].status == 5 and all(prog[i].status == 0 for i in (0, 1, 3, 4))
    assert (res[2]["depth"] == -7.0).all()
    for i in (0, 1, 3, 4):
        assert np.array_equal(res[i]["depth"], base[i]["depth"]) and np.array_equal(res[i]["conf"], base[i]["conf"])
        assert prog[i].filled == int((res[i]["conf"] > 0).sum())          # Progress::filled is per view
    # (b) a non-positive pixel footprint: the reference throws std::out_of_range for that view (fault injection:
    # with valid cameras the condition cannot be produced from outside)
    api.debug_inject_footprint(4)
    out = gpu_ctx.alloc_outputs(st, [0, 4, 1])
    out[1]["depth"].fill(-7.0)
    res = gpu_ctx.reconstruct(st, [0, 4, 1], out=out)
    assert [r["status"] for r in res] == [0, api.E_FOOTPRINT, 0]
    assert (res[1]["depth"] == -7.0).all()
    assert np.array_equal(res[0]["depth"], base[0]["depth"]) and np.array_equal(res[2]["depth"], base[1]["depth"])
    with pytest.raises(IndexError, match="Negative pixel footprint"):      # alone it is the call's outcome
        gpu_ctx.reconstruct(st, [4])
    api.debug_inject_footprint(-1)


def test_maps_vs_reference_scale0(ctx_g1, g1):
    r = ctx_g1.reconstruct(api.Settings(refViewNr=0), [0])[0]
    assert_map_parity(map_parity(r["depth"], r["conf"], g1["s0v0_depth"], g1["s0v0_conf"]))
    both = (r["depth"] > 0) & (g1["s0v0_depth"] > 0)
    assert np.abs(r["dz"][both] - g1["s0v0_dz"][both]).max() < 0.05
    # unfilled pixels are exactly zero in every map; the 2-pixel border is never filled (Q11)
    un = r["conf"] == 0
    assert (r["depth"][un] == 0).all() and (r["dz"][un] == 0).all() and (r["normal"][un] == 0).all()
    assert (r["depth"][:2] == 0).all() and (r["depth"][-2:] == 0).all()
    assert (r["depth"][:, :2] == 0).all() and (r["depth"][:, -2:] == 0).all()
    assert r["conf"].max() <= 1.0 and (r["conf"][~un] > 0).all()
    nrm = np.linalg.norm(r["normal"][~un], axis=1)
    assert np.abs(nrm - 1).max() < 1e-4


def test_maps_vs_reference_scale1_odd(gpu_ctx, g1b, g1b_scene):
    gpu_ctx.load_scene(g1b_scene)
    r = gpu_ctx.reconstruct(api.Settings(refViewNr=2, scale=1), [2])[0]
    assert r["depth"].shape == g1b["s1v2_depth"].shape
    # relative depth p99 on this fixture (161 x 120 at scale 1): the reference ALGORITHM against itself under six other queue
    # orders (the restatement, ORC_QUEUE_ORDER = reverse / random:1-3 / jitter:1-2) reaches 4.2e-3 ... 6.1e-3 -- the general
    # 5e-3 sits inside that spread here.  Measured (round 6): 5.04e-3; bound at 1.2 x that -- a regression that doubles the depth
    # error does not pass (round 5 had 9e-3 = 1.5 x the worst order probe here)
    m = map_parity(r["depth"], r["conf"], g1b["s1v2_depth"], g1b["s1v2_conf"])
    if os.environ.get("MI_TEST_PRINT"):
        print("G1b scale 1 view 2:", m)
    assert_map_parity(m, rel_p99=6.2e-3)


def test_two_views_local_neighbors_1(gpu_ctx, g2, g2_scene):
    gpu_ctx.load_scene(g2_scene)
    r = gpu_ctx.reconstruct(api.Settings(refViewNr=0, nrReconNeighbors=1), [0])[0]
    m = map_parity(r["depth"], r["conf"], g2["s0v0_depth"], g2["s0v0_conf"])
    # with a single neighbour the confidence is one NCC, not a mean of four: its order
    # sensitivity is about twice as large, so the p99 bound on conf is 1e-2 here
    assert m["iou"] >= 0.98 and m["rel_med"] <= 1e-3 and m["rel_p99"] <= 5e-3, m
    assert m["conf_med"] <= 1e-3 and m["conf_p99"] <= 1e-2, m
    assert m["conf_p99"] <= 7.9e-3, m                                 # (regression guard: 1.25 x the value measured in round 6)


def test_batch_equals_single_and_is_deterministic(ctx_g1, g1_scene):
    ctx_g1.load_scene(g1_scene)
    st = api.Settings()
    batch = ctx_g1.reconstruct(st, [0, 1, 2, 3, 4], want_views=True)
    again = ctx_g1.reconstruct(st, [0, 1, 2, 3, 4], want_views=True)
    stats = dict(ctx_g1.last_stats)
    for a, b in zip(batch, again):
        for k in ("depth", "conf", "dz", "normal", "views"):
            assert np.array_equal(a[k], b[k]), "run-to-run difference in %s" % k
    single = ctx_g1.reconstruct(st, [3], want_views=True)[0]
    for k in ("depth", "conf", "dz", "normal", "views"):
        assert np.array_equal(single[k], batch[3][k]), "batching changed %s" % k
    v = batch[0]["views"]
    filled = batch[0]["conf"] > 0
    assert (v[filled] >= 0).all() and (v[~filled] == -1).all()
    assert (np.diff(v[filled], axis=1) > 0).all()            # ascending ids (std::set order)
    assert stats["n_filled"] == sum(int((b["conf"] > 0).sum()) for b in again)


def test_every_view_vs_oracle(ctx_g1, g1_scene):
    from oracle import oracle as orc
    ctx_g1.load_scene(g1_scene)
    S = orc.OracleScene(g1_scene)
    res = ctx_g1.reconstruct(api.Settings(), [1, 2, 3, 4])
    for ref, r in zip([1, 2, 3, 4], res):
        o = S.reconstruct(orc.make_settings(ref_view=ref))
        assert_map_parity(map_parity(r["depth"], r["conf"], o["depth"], o["conf"]))


def test_dmrecon_class_drop_in(ctx_g1, g1_scene, g1):
    ctx_g1.load_scene(g1_scene)
    st = api.Settings(refViewNr=0, keepDzMap=True, keepConfidenceMap=True)
    recon = api.DMRecon(ctx_g1, st)
    assert recon.getRefViewNr() == 0
    recon.start()
    assert set(recon.images) == {"depth-L0", "dz-L0", "conf-L0"}            # dmrecon.cc:119-140
    assert recon.getProgress().status == 0                                   # RECON_IDLE
    assert recon.getProgress().filled == int((recon.images["depth-L0"] > 0).sum())
    assert_map_parity(map_parity(recon.images["depth-L0"], recon.images["conf-L0"], g1["s0v0_depth"], g1["s0v0_conf"]))
    st2 = api.Settings(refViewNr=1)
    r2 = api.DMRecon(ctx_g1, st2)
    r2.start()
    assert set(r2.images) == {"depth-L0"}


def test_edge_cases(gpu_ctx, g1_scene):
    from mve_amd.scene_io import SceneData
    # no features at all: nothing to seed, GVS finds nothing -> "Global View Selection failed"
    gpu_ctx.load_scene(SceneData(g1_scene.cameras, g1_scene.images, []))
    with pytest.raises(RuntimeError, match="Global View Selection failed"):
        gpu_ctx.reconstruct(api.Settings(), [0])
    gpu_ctx.load_scene(g1_scene)
    with pytest.raises(ValueError):
        gpu_ctx.reconstruct(api.Settings(), [9])                             # master view out of bounds
    with pytest.raises(ValueError):
        gpu_ctx.reconstruct(api.Settings(scale=-1), [0])
    with pytest.raises(ValueError):
        gpu_ctx.reconstruct(api.Settings(scale=12), [0])
    with pytest.raises(ValueError):
        gpu_ctx.reconstruct(api.Settings(filterWidth=4), [0])
    with pytest.raises(ValueError):
        gpu_ctx.reconstruct(api.Settings(filterWidth=13), [0])                # compiled for the odd widths 3..11
    with pytest.raises(ValueError):
        gpu_ctx.reconstruct(api.Settings(nrReconNeighbors=17), [0])      # more than MI_DMRECON_MAX_LOCAL_VIEWS
    with pytest.raises(ValueError):
        gpu_ctx.reconstruct(api.Settings(globalVSMax=129), [0])          # more than MI_DMRECON_MAX_GLOBAL_VIEWS
    # more local neighbours than the scene has other views: no patch finds them, nothing is filled (the reference
    # needs exactly K selected views, local_view_selection.cc:144-146), the call itself succeeds
    r = gpu_ctx.reconstruct(api.Settings(nrReconNeighbors=5), [0], want_views=True)[0]
    assert not (r["depth"] > 0).any() and r["views"].shape[2] == 8
    # an AABB that excludes every feature -> no global views
    with pytest.raises(RuntimeError, match="Global View Selection failed"):
        gpu_ctx.reconstruct(api.Settings(aabbMin=[100, 100, 100], aabbMax=[101, 101, 101]), [0])
    # in a batch the failing view is reported per view and the others still run
    # the coarsest level of 160x120 is 20x15 (level 3): almost no patch fits, the call still succeeds
    r = gpu_ctx.reconstruct(api.Settings(scale=3), [0])[0]
    assert r["depth"].shape == (15, 20)
    assert (r["depth"][:2] == 0).all() and (r["depth"][:, :2] == 0).all()
    # cancellation before start: nothing written (dmrecon.cc:101-105)
    prog = (api.CProgress * 1)()
    prog[0].cancelled = 1
    with pytest.raises(InterruptedError):
        gpu_ctx.reconstruct(api.Settings(), [0], progress=prog)
    assert prog[0].status == 5                                               # RECON_CANCELLED


# ---- scene H1: depth step, occluder, textureless band, a low-overlap view (tests/golden/make_golden_hard.py) -----

def test_hard_scene_patch_vectors_vs_reference(gpu_ctx, h1, h1_scene):
    """260 hypotheses dumped from the reference's own PatchOptimization on the hard scene: about half fail (samples
    leave a neighbour image, a propagated view is occluded and no replacement is good enough), propagated local
    sets get re-selected.  Same outcome class, depth / confidence within the patch-level tolerance, same views."""
    gpu_ctx.load_scene(h1_scene)
    assert gpu_ctx.global_view_selection(api.Settings(refViewNr=0)) == list(h1["gvs"])
    for lpv in (1, 16):
        out, loc = gpu_ctx.patch_optimize(api.Settings(refViewNr=0), 0, h1["seeds_xy"], h1["seeds_hyp"], h1["seeds_local"],
                                          lanes_per_view=lpv)
        ref, ref_loc = h1["opt"], h1["opt_local"]
        agree = (out[:, 0] > 0) == (ref[:, 0] > 0)
        ok = (out[:, 0] > 0) & (ref[:, 0] > 0)
        rel = np.abs(out[ok, 1] - ref[ok, 1]) / ref[ok, 1]
        dconf = np.abs(out[ok, 0] - ref[ok, 0])
        same_views = (loc[ok] == ref_loc[ok]).all(1)
        print("H1 patches lpv=%s: outcome agreement %.4f, ok %d, rel depth <=1e-3: %.4f, |dconf| <= 5e-3: %.4f, same views %.4f"
              % (lpv, agree.mean(), ok.sum(), (rel <= 1e-3).mean(), (dconf <= 5e-3).mean(), same_views.mean()))
        assert agree.mean() >= 0.97 and ok.sum() >= 100
        assert (rel <= 1e-3).mean() >= 0.97 and (dconf <= 5e-3).mean() >= 0.97 and same_views.mean() >= 0.97
        n = len(ref)
        changed = (ref[n // 2:, 0] > 0) & (ref_loc[n // 2:] != h1["seeds_local"][n // 2:]).any(1)
        both = changed & (out[n // 2:, 0] > 0)
        assert both.sum() >= 3 and (loc[n // 2:][both] == ref_loc[n // 2:][both]).all()      # re-selected like the reference


def test_hard_scene_maps_vs_reference(gpu_ctx, h1, h1_scene):
    """Map-level parity where the propagation meets discontinuities and empty regions.  Which neighbour reaches a
    pixel first decides its local view set, and at the depth step / the occluder's edge that decides the result:
    the reference algorithm against ITSELF with the queue popped worst-first (tests/test_oracle_golden.py::
    test_order_sensitivity_floor_on_hard_scene) gives IoU 0.9947, relative depth p99 1.6e-2, confidence p99 0.091 on
    view 0.  The bounds here are that floor x ~2 in the tails; medians as everywhere else."""
    gpu_ctx.load_scene(h1_scene)
    res = gpu_ctx.reconstruct(api.Settings(), [0, 8], want_views=True)
    stats = dict(gpu_ctx.last_stats)
    for v, r in zip((0, 8), res):
        m = map_parity(r["depth"], r["conf"], h1["s0v%d_depth" % v], h1["s0v%d_conf" % v])
        print("H1 view %d:" % v, m)
        assert m["iou"] >= 0.985 and m["rel_med"] <= 1e-3 and m["rel_p99"] <= 3e-2, m
        assert m["conf_med"] <= 2e-3 and m["conf_p99"] <= 0.15, m
        # regression guard (round 6): 1.25 x what this build measures here -- a view's maps are deterministic, so a change that moves
        # them further from the reference shows long before the floor-derived bounds above
        assert m["iou"] >= 0.997 and m["rel_p99"] <= 1.4e-2 and m["conf_med"] <= 5.5e-4 and m["conf_p99"] <= 7.5e-2, m
        # the occluder's silhouette and the empty part of the low-overlap view are where the reference has them
        empty_ref = h1["s0v%d_depth" % v] == 0
        assert ((r["depth"] == 0) & empty_ref).sum() >= 0.9 * empty_ref.sum()
    # the paths the smooth scenes never take were taken
    print("H1 stats:", {k: stats[k] for k in ("n_patch", "n_eval", "n_view_replaced", "n_iter14", "n_filled")})
    assert stats["n_view_replaced"] > 50 and stats["n_iter14"] >= 1


def test_views_registered_without_pixels(gpu_ctx, g1_scene):
    """mi_dmrecon_set_view(..., pixels = NULL): a view with its camera and image size only -- what the reference has for a view
    whose image cannot be loaded: still a candidate of the global view selections (dmrecon.cc:62-79), fatal only for the
    reconstructions that select it (dmrecon.cc:236-240: MI_DMRECON_ENOIMAGE here, per view), "Invalid master view" as a
    reference view itself."""
    gpu_ctx.load_scene(g1_scene)
    st = api.Settings(globalVSMax=2)
    sel = gpu_ctx.global_view_selection(st, 0)
    assert len(sel) == 2
    other = [v for v in range(1, 5) if v not in sel][0]
    ref = gpu_ctx.reconstruct(st, [0, other], want_views=True)
    h, w = g1_scene.images[0].shape[:2]
    # a view nobody of the call selects loses its pixels: same selection, same maps
    lost = [v for v in range(1, 5) if v not in sel and v not in gpu_ctx.global_view_selection(st, other) and v != other]
    if lost:
        gpu_ctx.set_view_camera_only(lost[0], g1_scene.cameras[lost[0]], w, h)
        assert gpu_ctx.global_view_selection(st, 0) == sel
        got = gpu_ctx.reconstruct(st, [0, other], want_views=True)
        for a, b in zip(got, ref):
            for k in ("depth", "conf", "dz", "views"):
                assert np.array_equal(a[k], b[k]), k
        with pytest.raises(ValueError, match="Invalid master view"):
            gpu_ctx.reconstruct(st, [lost[0]])
    # a view that view 0 selects loses its pixels: view 0 ends with E_NOIMAGE, the other view of the call finishes
    gpu_ctx.set_view_camera_only(sel[0], g1_scene.cameras[sel[0]], w, h)
    assert gpu_ctx.global_view_selection(st, 0) == sel                     # still a candidate, still selected
    if sel[0] not in gpu_ctx.global_view_selection(st, other):
        got = gpu_ctx.reconstruct(st, [0, other], want_views=True)
        assert got[0]["status"] == api.E_NOIMAGE and got[1]["status"] == 0
        assert np.array_equal(got[1]["depth"], ref[1]["depth"])
    with pytest.raises(RuntimeError, match="has no image"):
        gpu_ctx.reconstruct(st, [0])
    gpu_ctx.load_scene(g1_scene)                                           # every view with its pixels again
    again = gpu_ctx.reconstruct(st, [0, other], want_views=True)
    assert np.array_equal(again[0]["depth"], ref[0]["depth"])


def test_seed_reoptimisation_round(gpu_ctx, g1, g1_scene, h1, h1_scene, monkeypatch):
    """The reference's seed semantics -- a seed's OWN pixel is pushed (dmrecon.cc:316-326), re-optimised from its converged
    state when popped, and propagates only if that strictly raised its confidence (:365-398) -- in its two forms: INSIDE the
    seed launch (the default, MI_DMRECON_SEED_REOPT=2: every seed that succeeds is optimised once more from its own result,
    the propagation starts with round 2) and as a round of its own (=1: round 1 re-optimises every pixel the seeds wrote).
    Same maps, bit for bit.  =0 is the sweep's former default -- every seed propagates at once --, another propagation that
    meets the same tolerances on these scenes (and fills more than the reference on others: scene W2 below)."""
    for scene, fix, views, bounds in ((g1_scene, g1, (0,), None), (h1_scene, h1, (0, 8), (0.985, 3e-2, 0.15))):
        gpu_ctx.load_scene(scene)
        st = api.Settings()
        fold = gpu_ctx.reconstruct(st, list(views), want_views=True)                  # the default
        sf = dict(gpu_ctx.last_stats)
        monkeypatch.setenv("MI_DMRECON_SEED_REOPT", "0")
        base = gpu_ctx.reconstruct(st, list(views), want_views=True)
        s0 = dict(gpu_ctx.last_stats)
        monkeypatch.setenv("MI_DMRECON_SEED_REOPT", "1")
        got = gpu_ctx.reconstruct(st, list(views), want_views=True)
        s1 = dict(gpu_ctx.last_stats)
        assert s1["n_seeds_ok"] == s0["n_seeds_ok"] == sf["n_seeds_ok"] and s1["n_seeds"] == s0["n_seeds"]
        for a, b in zip(fold, got):
            for k in ("depth", "conf", "dz", "normal", "views"):
                assert np.array_equal(a[k], b[k]), k                                   # in the seed launch = as a round of its own
        assert sf["n_filled"] == s1["n_filled"] and sf["n_rounds"] == s1["n_rounds"] and sf["n_launches"] < s1["n_launches"]
        # (the seed launch re-optimises every seed that succeeds, the extra round only the one that won its pixel)
        assert s1["n_patch"] <= sf["n_patch"] <= s1["n_patch"] + s1["n_seeds_ok"], (sf["n_patch"], s1["n_patch"])
        for v, r, b in zip(views, got, base):
            m = map_parity(r["depth"], r["conf"], fix["s0v%d_depth" % v], fix["s0v%d_conf" % v])
            mb = map_parity(b["depth"], b["conf"], fix["s0v%d_depth" % v], fix["s0v%d_conf" % v])
            print("seed re-optimisation, view %d: with %s / without %s" % (v, {k: round(x, 5) for k, x in m.items()}, {k: round(x, 5) for k, x in mb.items()}))
            for mm in (m, mb):
                if bounds is None:
                    assert_map_parity(mm)
                else:
                    assert mm["iou"] >= bounds[0] and mm["rel_med"] <= 1e-3 and mm["rel_p99"] <= bounds[1] and mm["conf_p99"] <= bounds[2], mm
            assert not np.array_equal(r["depth"], b["depth"])            # it IS another propagation
        # the forms of a round with the extra round: plain launches / one launch / speculative, host-visible rounds only
        for env in ({"MI_DMRECON_ONE_LAUNCH": "0", "MI_DMRECON_SPEC_ROUNDS": "0"}, {"MI_DMRECON_SPEC_ROUNDS": "0"},
                    {"MI_DMRECON_SPEC_ROUNDS": "1000000"}, {"MI_DMRECON_ONE_LAUNCH": "0", "MI_DMRECON_SPEC_ROUNDS": "0", "MI_DMRECON_SINGLE_FOLLOW": "0"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            two = gpu_ctx.reconstruct(st, list(views), want_views=True)
            s2 = dict(gpu_ctx.last_stats)
            for k in env:
                monkeypatch.delenv(k)
            for a, b in zip(two, got):
                for k in ("depth", "conf", "dz", "normal", "views"):
                    assert np.array_equal(a[k], b[k]), (k, env)
            for k in ("n_patch", "n_eval", "n_filled", "n_rounds"):
                assert s2[k] == s1[k], (k, env, s2[k], s1[k])
        monkeypatch.delenv("MI_DMRECON_SEED_REOPT")
    gpu_ctx.load_scene(g1_scene)


# ---- apps/dmrecon --filter-width: 3 x 3, 7 x 7, 9 x 9 and 11 x 11 windows ------------------------------------

@pytest.mark.parametrize("fw", [3, 7, 9, 11])
def test_filter_widths_vs_reference(gpu_ctx, g1, g1_fw, g1_scene, fw, monkeypatch):
    """mvs::Settings::filterWidth 3, 7, 9 and 11 (the kernels are compiled once per width; from 9 x 9 on a lane of the
    latency layout holds two samples of the window): maps against the reference's own
    output of `dmrecon --filter-width=N`, patch results against its PatchOptimization class and against the oracle,
    in both lane layouts.  (Width 3: the reference's derivative step reads out of bounds, see tests/golden/
    make_golden_fw.py -- its patch results carry that noise, so the patch-level check is against the oracle there.)"""
    from oracle import oracle as orc
    gpu_ctx.load_scene(g1_scene)
    st = api.Settings(refViewNr=0, filterWidth=fw)
    r = gpu_ctx.reconstruct(st, [0])[0]
    m = map_parity(r["depth"], r["conf"], g1_fw["fw%d_depth" % fw], g1_fw["fw%d_conf" % fw])
    print("filter width %d maps:" % fw, m)
    assert m["iou"] >= 0.98 and m["rel_med"] <= 1e-3 and m["rel_p99"] <= 5e-3 and m["conf_med"] <= 1e-3, m
    assert m["conf_p99"] <= (2e-2 if fw == 3 else 5e-3), m
    assert m["conf_p99"] <= (1.7e-2 if fw == 3 else 5e-3), m          # (width 3: regression guard, 1.25 x the value measured in round 6)
    half = fw // 2                                                           # the border that is never filled (Q11)
    assert (r["depth"][:half] == 0).all() and (r["depth"][-half:] == 0).all() and (r["depth"][:, :half] == 0).all()
    S = orc.OracleScene(g1_scene)
    rng = np.random.RandomState(21)
    n = 300
    xy = np.stack([rng.randint(0, 160, n), rng.randint(0, 120, n)], 1)
    hyp = np.stack([10.0 + rng.uniform(-0.4, 0.4, n), rng.uniform(-5e-3, 5e-3, n), rng.uniform(-5e-3, 5e-3, n)], 1)
    oo, ol = S.patch_optimize(orc.make_settings(ref_view=0, filterWidth=fw), xy, hyp)
    for lpv in (1, 16):
        go, gl = gpu_ctx.patch_optimize(st, 0, xy, hyp, lanes_per_view=lpv)
        assert ((go[:, 0] > 0) == (oo[:, 0] > 0)).mean() >= 0.97
        ok = (go[:, 0] > 0) & (oo[:, 0] > 0)
        assert ok.sum() > 40
        # nine samples per view constrain a patch less than 25 or 49 do: the 3 x 3 optimisation is the most
        # sensitive to rounding (measured: 98.9 % of the patches within 1e-3)
        assert (np.abs(go[ok, 1] - oo[ok, 1]) / oo[ok, 1] <= 1e-3).mean() >= (0.97 if fw == 3 else 0.99)
        assert (np.abs(go[ok, 0] - oo[ok, 0]) <= 5e-3).mean() >= (0.96 if fw == 3 else 0.98)
        assert (gl[ok] == ol[ok]).all(1).mean() >= 0.98
        if fw >= 7:
            g2, g2l = gpu_ctx.patch_optimize(st, 0, g1["seeds_xy"], g1["seeds_hyp"], g1["seeds_local"], lanes_per_view=lpv)
            ref = g1_fw["fw%d_opt" % fw]
            both = (g2[:, 0] > 0) & (ref[:, 0] > 0)
            assert both.sum() >= 15 and ((g2[:, 0] > 0) == (ref[:, 0] > 0)).mean() >= 0.95
            assert (np.abs(g2[both, 1] - ref[both, 1]) / ref[both, 1] <= 1e-3).all()
    # the evaluation hook returns fw * fw samples per view
    e = gpu_ctx.patch_eval(st, 0, 80, 60, 10.0)
    o = S.patch_eval(orc.make_settings(ref_view=0, filterWidth=fw), 80, 60, 10.0)
    assert e["col"].shape[1] == fw * fw and np.array_equal(e["ok"], o["ok"])
    okv = e["ok"] > 0
    assert np.abs(e["col"][okv] - o["col"][okv]).max() <= 3e-5 and np.abs(e["ncc"][okv] - o["ncc"][okv]).max() <= 1e-4
    dref = o["deriv"][okv]
    assert np.abs(e["deriv"][okv] - dref).max() <= 1e-4 * max(np.abs(dref).max(), 1.0)


# ---- scene W1: 42 views -- more than 32 global views, more than four local views (tests/golden/make_golden_wide.py) ----

def test_wide_view_sets_vs_reference(gpu_ctx, w1, w1_scene, g1_scene, monkeypatch):
    """apps/dmrecon -n 40 --local-neighbors=6 / --local-neighbors=8 against the reference's own output: the global view
    selection of 40 views, maps reconstructed with six and eight local views per patch (the eight-slot lane layouts:
    an octet of lanes per patch in the throughput layout, eight 8-lane half rows in the latency layout), and the
    reference's own PatchOptimization on 160 hypotheses, half of them with a propagated set of six."""
    gpu_ctx.load_scene(w1_scene)
    st6 = api.Settings(refViewNr=0, nrReconNeighbors=6, globalVSMax=40)
    assert gpu_ctx.global_view_selection(st6) == list(w1["gvs40"]) and len(w1["gvs40"]) == 40
    for tag, st in (("k6n40", st6), ("k8n20", api.Settings(refViewNr=0, nrReconNeighbors=8, globalVSMax=20))):
        r = gpu_ctx.reconstruct(st, [0], want_views=True)[0]
        m = map_parity(r["depth"], r["conf"], w1[tag + "_depth"], w1[tag + "_conf"])
        # 112 x 84 images, 40 near-by views: the reference ALGORITHM is order-sensitive to rel_p99 8.3e-3 / conf median
        # 1.5e-2 / conf p99 0.12 here (its restatement with the queue popped worst-first against itself,
        # tests/test_oracle_golden.py::test_order_sensitivity_floor_on_wide_scene); bounds at ~1.5 x that floor,
        # the fill mask at the smooth-scene bound
        assert m["iou"] >= 0.99 and m["rel_med"] <= 1.5e-3 and m["rel_p99"] <= 1.3e-2, (tag, m)
        assert m["conf_med"] <= 2.2e-2 and m["conf_p99"] <= 0.18, (tag, m)
        # regression guard (round 6): 1.25 x what this build measures here -- a view's maps are deterministic, so a change that moves
        # them further from the reference shows long before the floor-derived bounds above
        assert m["rel_med"] <= 4.3e-4 and m["rel_p99"] <= 9.6e-3 and m["conf_med"] <= 2.8e-3 and m["conf_p99"] <= 0.126, (tag, m)
        filled = r["conf"] > 0
        v = r["views"][filled]
        assert v.shape[1] == 8 and ((v >= 0).sum(1) == st.nrReconNeighbors).all()           # exactly K views, ...
        k = st.nrReconNeighbors
        assert (np.diff(v[:, :k], axis=1) > 0).all() and (v[:, k:] == -1).all() and not (v == 0).any()   # ascending, never the reference view
        both = filled & (w1[tag + "_depth"] > 0)
        assert np.percentile(np.abs(r["dz"][both] - w1[tag + "_dz"][both]), 99) < 0.05
    # patch level, both lane layouts
    ref, ref_loc = w1["opt"], w1["opt_local"]
    for lpv in (1, 16):
        out, loc = gpu_ctx.patch_optimize(st6, 0, w1["seeds_xy"], w1["seeds_hyp"], w1["seeds_local"], lanes_per_view=lpv)
        assert loc.shape == (160, 8)
        assert ((out[:, 0] > 0) == (ref[:, 0] > 0)).mean() >= 0.97
        ok = (out[:, 0] > 0) & (ref[:, 0] > 0)
        assert ok.sum() >= 100
        # the patch-level bounds of the hard scene H1 (six views on 112 x 84 images: now and then a patch takes one
        # Gauss-Newton step more or fewer than the reference's build of the same arithmetic)
        rel, dconf = np.abs(out[ok, 1] - ref[ok, 1]) / ref[ok, 1], np.abs(out[ok, 0] - ref[ok, 0])
        print("W1 patches lpv=%d: ok %d, rel depth <= 1e-3: %.4f, |dconf| <= 5e-3: %.4f, same views %.4f"
              % (lpv, ok.sum(), (rel <= 1e-3).mean(), (dconf <= 5e-3).mean(), (loc[ok] == ref_loc[ok]).all(1).mean()))
        assert (rel <= 1e-3).mean() >= 0.97 and (dconf <= 5e-3).mean() >= 0.97
        assert (loc[ok] == ref_loc[ok]).all(1).mean() >= 0.97
    # three reference views in one call: the fused tail rounds and the front kernel with eight view slots write what
    # host-visible rounds in the same lane layout write
    refs = [0, 5, 11]
    monkeypatch.setenv("MI_DMRECON_FRONT", "0")
    monkeypatch.setenv("MI_DMRECON_VIEW_HANDOVER", "1000000000")
    monkeypatch.setenv("MI_DMRECON_HOST_ROUNDS", "1")
    seq = gpu_ctx.reconstruct(st6, refs, want_views=True)
    monkeypatch.delenv("MI_DMRECON_HOST_ROUNDS")
    for front in ("0", "1000000"):
        monkeypatch.setenv("MI_DMRECON_FRONT", front)
        got = gpu_ctx.reconstruct(st6, refs, want_views=True)
        assert gpu_ctx.last_stats["n_front_launches"] == (1 if front != "0" else 0)
        for a, b in zip(seq, got):
            for key in ("depth", "conf", "dz", "normal", "views"):
                assert np.array_equal(a[key], b[key]), (front, key)
    monkeypatch.delenv("MI_DMRECON_FRONT"); monkeypatch.delenv("MI_DMRECON_VIEW_HANDOVER")
    gpu_ctx.load_scene(g1_scene)                             # the scene the module's other tests expect (ctx_g1)


# ---- scene W2: 100 views -- more than 64 global views, more than eight local views (tests/golden/make_golden_wide2.py) ----

def test_wider_view_sets_vs_reference(gpu_ctx, w2, w2_scene, g1_scene, monkeypatch):
    """apps/dmrecon -n 80 --local-neighbors=10, --local-neighbors=16 and -n 80 with the default four local views against
    the reference's own output (the reference accepts any value of either, libs/dmrecon/settings.h:37-38): the global view
    selection of 80 views on the host and on the device, a second word of the availability mask and the larger NCC table of
    the view selection in every lane layout (K = 4: quads, fused tail rounds, front kernel), maps with ten and sixteen local
    views per patch (the sixteen-slot lane layout: a row of 16 lanes per patch, host-visible rounds only), and the
    reference's own PatchOptimization on 160 hypotheses, half of them with a propagated set of ten."""
    gpu_ctx.load_scene(w2_scene)
    st10 = api.Settings(refViewNr=0, nrReconNeighbors=10, globalVSMax=80)
    assert gpu_ctx.global_view_selection(st10) == list(w2["gvs80"]) and len(w2["gvs80"]) == 80
    monkeypatch.setenv("MI_DMRECON_GVS_DEVICE", "1")
    assert gpu_ctx.global_view_selection(st10) == list(w2["gvs80"])
    monkeypatch.delenv("MI_DMRECON_GVS_DEVICE")
    from oracle import oracle as orc
    S = orc.OracleScene(w2_scene)
    for tag, st in (("k10n80", st10), ("k16n20", api.Settings(refViewNr=0, nrReconNeighbors=16, globalVSMax=20)),
                    ("k4n80", api.Settings(refViewNr=0, nrReconNeighbors=4, globalVSMax=80))):
        # 96 x 72 images, up to 80 near-by views: the reference ALGORITHM against itself under four other queue orders
        # (its restatement with ORC_QUEUE_ORDER = reverse / random:1 / random:2 / jitter:1, measured when the fixture was
        # made) reaches, at worst: k10n80 IoU 1.0, rel_med 7.2e-4, rel_p99 8.3e-3, conf_med 1.4e-2, conf_p99 0.106; k16n20
        # IoU 0.9977, 3.1e-4, 3.2e-3, 2.3e-3, 0.048; k4n80 IoU 0.9960, 1.08e-3, 1.0e-2, 1.7e-2, 0.134 -- bounds at ~1.5 x the
        # worst of them.
        r = gpu_ctx.reconstruct(st, [0], want_views=True)[0]
        m = map_parity(r["depth"], r["conf"], w2[tag + "_depth"], w2[tag + "_conf"])
        print("W2", tag, m)
        assert m["iou"] >= 0.99 and m["rel_med"] <= 1.6e-3 and m["rel_p99"] <= 1.5e-2, (tag, m)
        assert m["conf_med"] <= 2.6e-2 and m["conf_p99"] <= 0.2, (tag, m)
        # regression guard (round 6): 1.25 x what this build measures here -- a view's maps are deterministic, so a change that moves
        # them further from the reference shows long before the floor-derived bounds above
        assert m["rel_med"] <= 4e-4 and m["rel_p99"] <= 1.15e-2 and m["conf_med"] <= 1.2e-3 and m["conf_p99"] <= 0.165, (tag, m)
        if tag == "k4n80":
            # The FILL MASK of this scene depends on the seed semantics ("Seeds"): a strip along the right
            # border is out of sight of the view sets that propagate towards it, so the region grown from the rest of the
            # image stops there in every queue order; the features INSIDE the strip select their own views, and in the
            # reference they only propagate if re-optimising them raises their confidence -- which is what the sweep does
            # now.  Its former default (MI_DMRECON_SEED_REOPT=0: every seed propagates at once) fills 3 % more pixels, every
            # one of them a patch the reference's own PatchOptimization accepts when given the same hypothesis and view set.
            monkeypatch.setenv("MI_DMRECON_SEED_REOPT", "0")
            r0 = gpu_ctx.reconstruct(st, [0], want_views=True)[0]
            monkeypatch.delenv("MI_DMRECON_SEED_REOPT")
            m0 = map_parity(r0["depth"], r0["conf"], w2[tag + "_depth"], w2[tag + "_conf"])
            print("W2", tag, "every seed propagates at once", m0)
            assert 0.96 <= m0["iou"] < m["iou"] and m0["rel_p99"] <= 1.5e-2 and m0["conf_p99"] <= 0.2, m0
            extra = (r0["depth"] > 0) & ~(w2[tag + "_depth"] > 0)
            ys, xs = np.nonzero(extra)
            out, oloc = S.patch_optimize(orc.make_settings(ref_view=0, local_neighbors=st.nrReconNeighbors, global_max=st.globalVSMax),
                                         np.stack([xs, ys], 1), np.stack([r0["depth"][extra], r0["dz"][extra][:, 0], r0["dz"][extra][:, 1]], 1),
                                         r0["views"][extra])
            print("W2", tag, "pixels only that form fills: %d, accepted by the reference's PatchOptimization: %d, |dconf| median %.1e"
                  % (extra.sum(), (out[:, 0] > 0).sum(), np.median(np.abs(out[:, 0] - r0["conf"][extra]))))
            assert extra.sum() > 100 and (out[:, 0] > 0).mean() >= 0.97 and np.median(np.abs(out[:, 0] - r0["conf"][extra])) <= 5e-3
        filled = r["conf"] > 0
        v = r["views"][filled]
        k = st.nrReconNeighbors
        assert v.shape[1] == api.local_view_channels(k) and ((v >= 0).sum(1) == k).all()          # exactly K views, ...
        assert (np.diff(v[:, :k], axis=1) > 0).all() and (v[:, k:] == -1).all() and not (v == 0).any()   # ascending, never the reference view
        assert set(np.unique(v[:, :k])) <= set(int(g) for g in gpu_ctx.global_view_selection(st))
        both = filled & (w2[tag + "_depth"] > 0)
        # (dz: the reference against itself under the other queue orders reaches p99 0.045 / 0.028 / 0.058 on the three settings)
        assert np.percentile(np.abs(r["dz"][both] - w2[tag + "_dz"][both]), 99) < 0.09, tag
    # local views with indices 64..79 of the global list (the second word of the availability mask, the upper half of the
    # NCC table) are in every set of ten and in a quarter of the sets of four; another reference view, here against the
    # restatement (bit-identical to the reference on this scene: tests/test_oracle_golden.py)
    for k in (4,):
        st = api.Settings(refViewNr=99, nrReconNeighbors=k, globalVSMax=80)
        r = gpu_ctx.reconstruct(st, [99], want_views=True)[0]
        o = S.reconstruct(orc.make_settings(ref_view=99, local_neighbors=k, global_max=80))
        m = map_parity(r["depth"], r["conf"], o["depth"], o["conf"])
        print("W2 view 99, K = %d" % k, m)
        assert m["iou"] >= 0.99 and m["rel_med"] <= 1.6e-3 and m["rel_p99"] <= 1.5e-2 and m["conf_med"] <= 2.6e-2 and m["conf_p99"] <= 0.2, (k, m)
        gl = np.asarray(gpu_ctx.global_view_selection(st))
        assert len(gl) == 80 and list(gl) == S.global_vs(orc.make_settings(ref_view=99, global_max=80))
        v = r["views"][r["conf"] > 0][:, :k]
        assert (np.searchsorted(gl, v) >= 64).any(1).mean() > 0.1, np.bincount(np.searchsorted(gl, v).ravel(), minlength=80)
    # patch level: ten local views, half of the hypotheses with a propagated set of ten (view slots 8, 9 travel in
    # DevJob::hyp_x / results_x)
    ref, ref_loc = w2["opt"], w2["opt_local"]
    out, loc = gpu_ctx.patch_optimize(st10, 0, w2["seeds_xy"], w2["seeds_hyp"], w2["seeds_local"], lanes_per_view=1)
    assert loc.shape == (160, 16)
    with pytest.raises(ValueError):                          # (more than eight local views: the throughput layout only)
        gpu_ctx.patch_optimize(st10, 0, w2["seeds_xy"], w2["seeds_hyp"], w2["seeds_local"], lanes_per_view=16)
    assert ((out[:, 0] > 0) == (ref[:, 0] > 0)).mean() >= 0.97
    ok = (out[:, 0] > 0) & (ref[:, 0] > 0)
    assert ok.sum() >= 100
    rel, dconf = np.abs(out[ok, 1] - ref[ok, 1]) / ref[ok, 1], np.abs(out[ok, 0] - ref[ok, 0])
    print("W2 patches: ok %d, rel depth <= 1e-3: %.4f, |dconf| <= 5e-3: %.4f, same views %.4f"
          % (ok.sum(), (rel <= 1e-3).mean(), (dconf <= 5e-3).mean(), (loc[ok] == ref_loc[ok]).all(1).mean()))
    assert (rel <= 1e-3).mean() >= 0.97 and (dconf <= 5e-3).mean() >= 0.97
    assert (loc[ok] == ref_loc[ok]).all(1).mean() >= 0.97
    # four local views out of 80 global ones, three reference views in one call: the fused tail rounds and the front kernel
    # write what host-visible rounds in the same lane layout write (the two-word availability mask in every kernel)
    st4 = api.Settings(refViewNr=0, nrReconNeighbors=4, globalVSMax=80)
    refs = [0, 41, 77]
    monkeypatch.setenv("MI_DMRECON_FRONT", "0")
    monkeypatch.setenv("MI_DMRECON_VIEW_HANDOVER", "1000000000")
    monkeypatch.setenv("MI_DMRECON_HOST_ROUNDS", "1")
    seq = gpu_ctx.reconstruct(st4, refs, want_views=True)
    monkeypatch.delenv("MI_DMRECON_HOST_ROUNDS")
    for front in ("0", "1000000"):
        monkeypatch.setenv("MI_DMRECON_FRONT", front)
        got = gpu_ctx.reconstruct(st4, refs, want_views=True)
        for a, b in zip(seq, got):
            for key in ("depth", "conf", "dz", "normal", "views"):
                assert np.array_equal(a[key], b[key]), (front, key)
    monkeypatch.delenv("MI_DMRECON_FRONT"); monkeypatch.delenv("MI_DMRECON_VIEW_HANDOVER")
    # sixteen slots: a batch of three views writes what the views write alone (nothing of a call's plan shows in a view's maps)
    alone = [gpu_ctx.reconstruct(st10, [v], want_views=True)[0] for v in refs]
    batch = gpu_ctx.reconstruct(st10, refs, want_views=True)
    for a, b in zip(alone, batch):
        for key in ("depth", "conf", "dz", "normal", "views"):
            assert np.array_equal(a[key], b[key]), key
    gpu_ctx.load_scene(g1_scene)                             # the scene the module's other tests expect (ctx_g1)
