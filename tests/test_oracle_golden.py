"""CPU tests: the oracle restatement (oracle/dmrecon_oracle.cc) against fixtures produced by the
REAL reference (tests/golden/make_golden.py ran oracle/_ref/dmrecon_ref_strict and
oracle/_ref/ref_patch_driver, i.e. the unmodified libs/dmrecon compiled -O2 -ffp-contract=off).

The restatement mirrors the reference's float accumulation order, so these are exact comparisons.
"""
import numpy as np
import pytest

from oracle import oracle as orc


def test_srgb_table_matches_reference_literal(g1):
    # the literal table of libs/dmrecon/mvs_tools.cc:30-93 vs the formula at :22-29 in double
    i = np.arange(256)
    x = i / 255.0
    y = np.where(i <= 0.04045 * 255.0, x / 12.92, ((x + 0.055) / 1.055) ** 2.4).astype(np.float32)
    y[255] = 1.0
    assert np.array_equal(y, g1["srgb2lin"])


def test_global_view_selection(g1, g1_scene):
    S = orc.OracleScene(g1_scene)
    assert S.global_vs(orc.make_settings(ref_view=0)) == list(g1["gvs"])


def test_full_maps_scale0_bit_exact(g1, g1_scene):
    S = orc.OracleScene(g1_scene)
    r = S.reconstruct(orc.make_settings(ref_view=0, scale=0))
    assert np.array_equal(r["depth"], g1["s0v0_depth"])
    assert np.array_equal(r["conf"], g1["s0v0_conf"])
    assert np.array_equal(r["dz"], g1["s0v0_dz"])
    assert r["stats"]["n_filled"] == int((g1["s0v0_depth"] > 0).sum())


def test_full_maps_scale1_odd_size_bit_exact(g1b, g1b_scene):
    # 322x241 -> 161x121: exercises the principal-point rescale on odd sizes (image_pyramid.cc:39-44)
    S = orc.OracleScene(g1b_scene)
    img, _, _ = S.pyramid_level(2, 1)
    assert np.array_equal(img, g1b["s1v2_undist"])          # byte-exact Gaussian pyramid level
    r = S.reconstruct(orc.make_settings(ref_view=2, scale=1))
    assert np.array_equal(r["depth"], g1b["s1v2_depth"])
    assert np.array_equal(r["conf"], g1b["s1v2_conf"])
    assert np.array_equal(r["dz"], g1b["s1v2_dz"])


def test_two_views_local_neighbors_1(g2, g2_scene):
    # BASELINE config 1 shape: 2 views need --local-neighbors=1
    S = orc.OracleScene(g2_scene)
    r = S.reconstruct(orc.make_settings(ref_view=0, scale=0, local_neighbors=1))
    assert np.array_equal(r["depth"], g2["s0v0_depth"])
    assert np.array_equal(r["conf"], g2["s0v0_conf"])
    r4 = S.reconstruct(orc.make_settings(ref_view=0, scale=0, local_neighbors=4))
    assert (r4["depth"] > 0).sum() == 0                      # default settings fill nothing with 2 views


def test_patch_optimization_vs_reference_classes(g1, g1_scene):
    S = orc.OracleScene(g1_scene)
    out, loc = S.patch_optimize(orc.make_settings(ref_view=0), g1["seeds_xy"], g1["seeds_hyp"], g1["seeds_local"])
    ref, ref_loc = g1["opt"], g1["opt_local"]
    assert np.array_equal(out[:, 0] > 0, ref[:, 0] > 0)
    assert (ref[:4, 0] == 0).all()                           # border patches fail
    ok = ref[:, 0] > 0
    assert ok.sum() >= 10
    # text round trip of the driver prints 9 significant digits -> exact float32
    assert np.array_equal(out[ok, :7], ref[ok, :7])
    assert np.array_equal(loc[ok], ref_loc[ok])


def test_patch_sampler_vs_reference_classes(g1, g1_scene):
    S = orc.OracleScene(g1_scene)
    st = orc.make_settings(ref_view=0)
    n_checked = 0
    for i in range(len(g1["seeds_xy"])):
        x, y = [int(v) for v in g1["seeds_xy"][i]]
        d, dzi, dzj = [float(v) for v in g1["seeds_hyp"][i]]
        e = S.patch_eval(st, x, y, d, dzi, dzj)
        assert e["master"][0] == g1["ev_master"][i, 0]
        if not e["master"][0]:
            continue
        assert np.array_equal(e["master"][1:], g1["ev_master"][i, 1:])
        assert np.array_equal(e["ncc"], g1["ev_ncc"][i])
        assert np.array_equal(e["ok"], g1["ev_ok"][i])
        okv = e["ok"] > 0
        assert np.array_equal(e["col"][okv], g1["ev_col"][i][okv])
        assert np.array_equal(e["deriv"][okv], g1["ev_der"][i][okv])
        n_checked += int(okv.sum())
    assert n_checked > 50


def test_hard_scene_maps_bit_exact(h1, h1_scene):
    """Scene H1 (depth step, occluder, textureless band, a low-overlap view): the restatement reproduces the
    reference's maps bit for bit also where views get replaced, samplings fail and regions stay empty."""
    S = orc.OracleScene(h1_scene)
    for v in (0, 8):
        r = S.reconstruct(orc.make_settings(ref_view=v, scale=0))
        assert np.array_equal(r["depth"], h1["s0v%d_depth" % v]), v
        assert np.array_equal(r["conf"], h1["s0v%d_conf" % v]), v
        assert np.array_equal(r["dz"], h1["s0v%d_dz" % v]), v
    assert 0.15 < (h1["s0v8_depth"] > 0).mean() < 0.35 and 0.6 < (h1["s0v0_depth"] > 0).mean() < 0.9


def test_hard_scene_patch_vectors_exact(h1, h1_scene):
    S = orc.OracleScene(h1_scene)
    assert S.global_vs(orc.make_settings(ref_view=0)) == list(h1["gvs"])
    out, loc = S.patch_optimize(orc.make_settings(ref_view=0), h1["seeds_xy"], h1["seeds_hyp"], h1["seeds_local"])
    ref, ref_loc = h1["opt"], h1["opt_local"]
    assert np.array_equal(out[:, 0] > 0, ref[:, 0] > 0)
    ok = ref[:, 0] > 0
    assert ok.sum() >= 100 and (~ok).sum() >= 100           # both outcomes are well represented
    assert np.array_equal(out[ok, :7], ref[ok, :7]) and np.array_equal(loc[ok], ref_loc[ok])
    n = len(ref)
    changed = ok[n // 2:] & (ref_loc[n // 2:] != h1["seeds_local"][n // 2:]).any(1)
    assert changed.sum() >= 3                                # propagated sets the reference had to re-select


def test_order_sensitivity_floor_on_hard_scene(h1, h1_scene, monkeypatch):
    """How much does the reference ALGORITHM depend on its own pop order?  Same restatement, queue popped worst-first
    (ORC_QUEUE_ORDER=reverse, read when the library is loaded -> a subprocess).  This is the floor against which the
    GPU's parallel sweep is judged on scene H1 (tests/test_gpu_parity.py::test_hard_scene_maps_vs_reference)."""
    import os
    import subprocess
    import sys
    import tempfile
    from conftest import map_parity, ROOT
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from conftest import scene_from_golden, GOLDEN\nfrom oracle import oracle as orc\nimport os\n"
            "g = dict(np.load(os.path.join(GOLDEN, 'h1_hard_9views_208x156.npz')))\n"
            "r = orc.OracleScene(scene_from_golden(g)).reconstruct(orc.make_settings(ref_view=0, scale=0))\n"
            "np.savez(sys.argv[1], d=r['depth'], c=r['conf'])\n" % (ROOT, os.path.join(ROOT, "tests")))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rev.npz")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, ORC_QUEUE_ORDER="reverse"))
        rev = np.load(out)
        m = map_parity(h1["s0v0_depth"], h1["s0v0_conf"], rev["d"], rev["c"])
    assert not np.array_equal(rev["d"], h1["s0v0_depth"])
    # measured: iou 0.9947, rel_med 2.5e-4, rel_p99 1.6e-2, conf_med 9.5e-4, conf_p99 0.091
    assert 0.99 <= m["iou"] <= 0.999 and 5e-3 <= m["rel_p99"] <= 3e-2 and 0.03 <= m["conf_p99"] <= 0.15, m


def test_filter_widths_7_9_11_bit_exact_and_3_to_rounding(g1, g1_fw, g1_scene):
    """apps/dmrecon --filter-width (mvs::Settings::filterWidth): 7 x 7 windows keep quirk Q3 (derivative step at
    patchPoints[12] = row 1, column 5) and are restated bit for bit.  With 3 x 3 windows the reference reads
    patchPoints[12] of a 9-element vector (undefined behaviour); the restatement uses the centre sample, the step
    only scales a finite difference that is divided out again: agreement to rounding."""
    from conftest import map_parity
    S = orc.OracleScene(g1_scene)
    for fw in (7, 9, 11):                                    # 9 x 9 and 11 x 11: the same quirk, the same bits
        r = S.reconstruct(orc.make_settings(ref_view=0, filterWidth=fw))
        assert np.array_equal(r["depth"], g1_fw["fw%d_depth" % fw]) and np.array_equal(r["conf"], g1_fw["fw%d_conf" % fw]), fw
        assert np.array_equal(r["dz"], g1_fw["fw%d_dz" % fw]), fw
        out, loc = S.patch_optimize(orc.make_settings(ref_view=0, filterWidth=fw), g1["seeds_xy"], g1["seeds_hyp"], g1["seeds_local"])
        ok = g1_fw["fw%d_opt" % fw][:, 0] > 0
        assert ok.sum() >= 15 and np.array_equal(out[:, 0] > 0, ok), fw
        assert np.array_equal(out[ok, :7], g1_fw["fw%d_opt" % fw][ok, :7]) and np.array_equal(loc[ok], g1_fw["fw%d_opt_local" % fw][ok]), fw
    for bad in (1, 4, 13):                                   # even widths are undefined behaviour in the reference
        with pytest.raises(Exception):
            S.reconstruct(orc.make_settings(ref_view=0, filterWidth=bad))
    r = S.reconstruct(orc.make_settings(ref_view=0, filterWidth=3))
    m = map_parity(r["depth"], r["conf"], g1_fw["fw3_depth"], g1_fw["fw3_conf"])
    assert m["iou"] >= 0.99 and m["rel_med"] <= 1e-3 and m["rel_p99"] <= 5e-3 and m["conf_p99"] <= 2e-2, m
    assert not np.array_equal(g1_fw["fw3_depth"], g1["s0v0_depth"]) and not np.array_equal(g1_fw["fw7_depth"], g1["s0v0_depth"])


def test_more_than_four_local_and_32_global_views(w1, w1_scene):
    """apps/dmrecon -n 40 --local-neighbors=6 and --local-neighbors=8 on scene W1 (42 views): the restatement is bit
    for bit the reference there too (global view selection of 40 views, local sets of six / eight)."""
    S = orc.OracleScene(w1_scene)
    st = orc.make_settings(ref_view=0, local_neighbors=6, global_max=40)
    assert S.global_vs(st) == list(w1["gvs40"]) and len(w1["gvs40"]) == 40
    r = S.reconstruct(st)
    assert np.array_equal(r["depth"], w1["k6n40_depth"]) and np.array_equal(r["conf"], w1["k6n40_conf"])
    assert np.array_equal(r["dz"], w1["k6n40_dz"])
    r = S.reconstruct(orc.make_settings(ref_view=0, local_neighbors=8, global_max=20))
    assert np.array_equal(r["depth"], w1["k8n20_depth"]) and np.array_equal(r["conf"], w1["k8n20_conf"])
    out, loc = S.patch_optimize(st, w1["seeds_xy"], w1["seeds_hyp"], w1["seeds_local"])
    ok = w1["opt"][:, 0] > 0
    assert ok.sum() >= 100 and np.array_equal(out[:, 0] > 0, ok)
    assert np.array_equal(out[ok, :7], w1["opt"][ok, :7]) and np.array_equal(loc[ok], w1["opt_local"][ok])
    assert ((w1["opt_local"][ok] >= 0).sum(1) == 6).all()


def test_more_than_eight_local_and_64_global_views(w2, w2_scene):
    """apps/dmrecon -n 80 --local-neighbors=10, --local-neighbors=16 and -n 80 alone on scene W2 (100 views): the reference
    accepts any number of either (libs/dmrecon/settings.h:37-38, local_view_selection.cc:144-146); the restatement is bit for
    bit the reference there too."""
    S = orc.OracleScene(w2_scene)
    st = orc.make_settings(ref_view=0, local_neighbors=10, global_max=80)
    assert S.global_vs(st) == list(w2["gvs80"]) and len(w2["gvs80"]) == 80
    for tag, k, ng in (("k10n80", 10, 80), ("k16n20", 16, 20), ("k4n80", 4, 80)):
        r = S.reconstruct(orc.make_settings(ref_view=0, local_neighbors=k, global_max=ng))
        assert np.array_equal(r["depth"], w2[tag + "_depth"]) and np.array_equal(r["conf"], w2[tag + "_conf"]), tag
        assert np.array_equal(r["dz"], w2[tag + "_dz"]), tag
    out, loc = S.patch_optimize(st, w2["seeds_xy"], w2["seeds_hyp"], w2["seeds_local"])
    ok = w2["opt"][:, 0] > 0
    assert ok.sum() >= 100 and np.array_equal(out[:, 0] > 0, ok) and loc.shape == (160, 16)
    assert np.array_equal(out[ok, :7], w2["opt"][ok, :7]) and np.array_equal(loc[ok], w2["opt_local"][ok])
    assert ((w2["opt_local"][ok] >= 0).sum(1) == 10).all()


def test_order_sensitivity_floor_on_wide_scene(w1):
    """The reference algorithm against itself with its queue popped worst-first on scene W1 (small images, 40 near-by
    global views, six local views): the floor behind the bounds of tests/test_gpu_parity.py::
    test_wide_view_sets_vs_reference."""
    import os
    import subprocess
    import sys
    import tempfile
    from conftest import map_parity, ROOT
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from conftest import scene_from_golden, GOLDEN\nfrom oracle import oracle as orc\nimport os\n"
            "g = dict(np.load(os.path.join(GOLDEN, 'w1_wide_42views_112x84.npz')))\n"
            "r = orc.OracleScene(scene_from_golden(g)).reconstruct(orc.make_settings(ref_view=0, local_neighbors=6, global_max=40))\n"
            "np.savez(sys.argv[1], d=r['depth'], c=r['conf'])\n" % (ROOT, os.path.join(ROOT, "tests")))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rev.npz")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, ORC_QUEUE_ORDER="reverse"))
        rev = np.load(out)
        m = map_parity(w1["k6n40_depth"], w1["k6n40_conf"], rev["d"], rev["c"])
    # measured: iou 1.0, rel_med 8.7e-4, rel_p99 8.3e-3, conf_med 1.5e-2, conf_p99 0.12
    assert m["iou"] >= 0.995 and 4e-3 <= m["rel_p99"] <= 1.5e-2 and 5e-3 <= m["conf_med"] <= 3e-2 and 0.06 <= m["conf_p99"] <= 0.2, m


def test_order_sensitivity_floors_behind_the_newer_bounds(w2, g1b):
    """The floors two bounds of the GPU tests are set against, measured again here: the reference algorithm (its restatement)
    against the reference binary's own maps with its queue popped worst-first (ORC_QUEUE_ORDER=reverse).  Scene W2 (100 views
    of 96 x 72; -n 80 with four and with ten local views): tests/test_gpu_parity.py::test_wider_view_sets_vs_reference.  Fixture
    G1b at scale 1 (161 x 120): the relative-depth p99 of ::test_maps_vs_reference_scale1_odd, where the general 5e-3 sits inside
    the algorithm's own spread."""
    import os
    import subprocess
    import sys
    import tempfile
    from conftest import map_parity, ROOT
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from conftest import scene_from_golden, GOLDEN\nfrom oracle import oracle as orc\nimport os\n"
            "g = dict(np.load(os.path.join(GOLDEN, 'w2_wider_100views_96x72.npz')))\n"
            "S = orc.OracleScene(scene_from_golden(g))\n"
            "a = S.reconstruct(orc.make_settings(ref_view=0, local_neighbors=4, global_max=80))\n"
            "b = S.reconstruct(orc.make_settings(ref_view=0, local_neighbors=10, global_max=80))\n"
            "g = dict(np.load(os.path.join(GOLDEN, 'g1b_5views_322x241_scale1.npz')))\n"
            "c = orc.OracleScene(scene_from_golden(g)).reconstruct(orc.make_settings(ref_view=2, scale=1))\n"
            "np.savez(sys.argv[1], ad=a['depth'], ac=a['conf'], az=a['dz'], bd=b['depth'], bc=b['conf'], bz=b['dz'], cd=c['depth'], cc=c['conf'])\n"
            % (ROOT, os.path.join(ROOT, "tests")))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rev.npz")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, ORC_QUEUE_ORDER="reverse"))
        rev = np.load(out)
        m4 = map_parity(rev["ad"], rev["ac"], w2["k4n80_depth"], w2["k4n80_conf"])
        m10 = map_parity(rev["bd"], rev["bc"], w2["k10n80_depth"], w2["k10n80_conf"])
        mg = map_parity(rev["cd"], rev["cc"], g1b["s1v2_depth"], g1b["s1v2_conf"])
        both = (rev["ad"] > 0) & (w2["k4n80_depth"] > 0)
        dz4 = float(np.percentile(np.abs(rev["az"][both] - w2["k4n80_dz"][both]), 99))
    # measured: W2 four local views: IoU 0.9960, rel_med 1.08e-3, rel_p99 9.6e-3, conf_med 1.7e-2, conf_p99 0.133, dz p99 0.058;
    # ten: IoU 1.0, 7.2e-4, 7.1e-3, 1.4e-2, 0.106; G1b scale 1: rel_p99 6.1e-3 (the other orders: 4.2e-3 ... 5.4e-3)
    assert m4["iou"] >= 0.99 and 5e-3 <= m4["rel_p99"] <= 1.5e-2 and 8e-3 <= m4["conf_med"] <= 2.6e-2 and 0.07 <= m4["conf_p99"] <= 0.2, m4
    assert 0.03 <= dz4 <= 0.09, dz4
    assert m10["iou"] >= 0.995 and 3.5e-3 <= m10["rel_p99"] <= 1.5e-2 and 0.05 <= m10["conf_p99"] <= 0.2, m10
    assert mg["iou"] >= 0.999 and 4e-3 <= mg["rel_p99"] <= 9e-3, mg


def test_queue_order_probes_are_orders_of_the_same_algorithm(g1, g1_scene, monkeypatch):
    """ORC_QUEUE_ORDER (read by the restatement at every reconstruction): unset -> the reference's own pop order, bit for bit
    (the fixtures above); reverse / random:<seed> / jitter:<seed> -> the same algorithm in another valid order: a probe of how
    much the result depends on the order (tools/order_floor.py -> tests/golden/order_floor_c3.json, the floor the fill-mask
    bounds of the GPU sweep are set against).  The probes are deterministic, differ from the reference order in the last digits
    only, and leave the default untouched."""
    import json
    import os
    S = orc.OracleScene(g1_scene)
    st = orc.make_settings(ref_view=0, scale=0)
    got = {}
    for order in ("reverse", "random:1", "random:2", "jitter:1"):
        monkeypatch.setenv("ORC_QUEUE_ORDER", order)
        a = S.reconstruct(st)
        b = S.reconstruct(st)
        assert np.array_equal(a["depth"], b["depth"])                    # deterministic
        got[order] = a
    monkeypatch.delenv("ORC_QUEUE_ORDER")
    ref = S.reconstruct(st)
    assert np.array_equal(ref["depth"], g1["s0v0_depth"])                # the default is still the reference's order
    for order, a in got.items():
        ma, mb = a["depth"] > 0, ref["depth"] > 0
        both = ma & mb
        assert both.sum() / max((ma | mb).sum(), 1) > 0.9, order
        rel = np.abs(a["depth"][both] - ref["depth"][both]) / ref["depth"][both]
        assert np.median(rel) < 5e-3 and not np.array_equal(a["depth"], ref["depth"]), order
    assert not np.array_equal(got["random:1"]["depth"], got["random:2"]["depth"])
    # the committed floor of the C3 scene: seven orders for every one of its 20 views
    f = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "order_floor_c3.json")))
    assert len(f["views"]) == 20 and all(len(v["orders"]) == 7 for v in f["views"].values())
    assert abs(f["min_fill_iou"] - min(f["worst"]["iou"].values())) < 1e-12 and 0.9 < f["min_fill_iou"] < 0.98
