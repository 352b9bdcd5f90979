"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mi_dmrecon.h declares, error behaviour without a GPU, and the host-side mirror of
mvs::Settings / mvs::DMRecon (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from mve_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(header="mi_dmrecon.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_dmrecon_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = api.load_library()
    names = header_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(L, n), "libmi_dmrecon.so does not export %s" % n
    assert sorted(api.EXPORTS) == names


def test_library_exports_nothing_undeclared():
    """... and nothing else: the dynamic symbols the library defines are exactly the functions of include/mi_dmrecon.h plus
    the test hooks of include/mi_dmrecon_debug.h (mve_amd/csrc/exports.map keeps launchers, kernel handles and C++ helpers
    local; a hook nobody declared, or a C++ symbol that slipped out, fails here)."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH], text=True)
    exported = sorted({l.split()[-1].split("@")[0] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] in "TDBRW"})
    hooks = header_functions("mi_dmrecon_debug.h")
    assert len(hooks) == 6 and all(h.startswith("mi_dmrecon_debug_") for h in hooks)
    assert exported == sorted(header_functions() + hooks), sorted(set(exported) ^ set(header_functions() + hooks))


def test_stats_struct_carries_its_size():
    """mi_dmrecon_stats::struct_size: the caller says how much room it has, the library never writes beyond it (a caller
    built against an older, shorter header) -- and an object whose size was never set is refused.  No GPU needed: the
    argument check comes first."""
    L = api.load_library()
    st = api.Settings().to_c()
    stats = api.CStats()                                       # struct_size = 0: never initialised
    refs = np.zeros(1, np.int32)
    maps = (api.CMaps * 1)()
    rc = L.mi_dmrecon_reconstruct(None, ctypes.byref(st), 1, refs.ctypes.data_as(ctypes.c_void_p), maps, None, None, ctypes.byref(stats))
    assert rc == api.E_INVAL and b"struct_size" in L.mi_dmrecon_last_error()
    # a short object (the first 16 fields of an "older header") followed by a canary: the canary survives the call
    buf = (ctypes.c_int64 * 32)(*([0x5A5A5A5A5A5A5A5A] * 32))
    buf[0] = 16 * 8
    rc = L.mi_dmrecon_reconstruct(None, ctypes.byref(st), 1, refs.ctypes.data_as(ctypes.c_void_p), maps, None, None,
                                  ctypes.cast(buf, ctypes.POINTER(api.CStats)))
    assert rc != 0                                             # (null context: the call itself fails ...)
    assert buf[0] == 16 * 8 and all(buf[i] == 0 for i in range(1, 16))       # ... its statistics are zeros, in the caller's size
    assert all(buf[i] == 0x5A5A5A5A5A5A5A5A for i in range(16, 32))


def header_struct(name):
    """[(ctype, field)] of a typedef struct in include/mi_dmrecon.h, arrays as 'type[n]'."""
    src = open(os.path.join(ROOT, "include", "mi_dmrecon.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(?:volatile )?(?:const )?([A-Za-z0-9_]+\s*\*?)\s*(.*)", decl)
        ctype = m.group(1).replace(" ", "")
        for f in m.group(2).split(","):
            f = f.strip()
            arr = re.match(r"([A-Za-z0-9_]+)\[(\d+)\]", f)
            out.append((ctype + ("[%s]" % arr.group(2) if arr else ""), arr.group(1) if arr else f.lstrip("*").strip()))
    return out


def test_ctypes_mirrors_match_the_header_field_for_field():
    """mve_amd/api.py restates the structs of include/mi_dmrecon.h for ctypes: same fields, same order, same types
    (a field added on one side only shifts everything behind it silently)."""
    C = {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "float": ctypes.c_float,
         "double": ctypes.c_double}
    for cname, mirror in (("mi_dmrecon_settings", api.CSettings), ("mi_dmrecon_stats", api.CStats)):
        want = header_struct(cname)
        got = mirror._fields_
        assert [f for _, f in want] == [f for f, _ in got], cname
        for (ct, f), (_, gt) in zip(want, got):
            arr = re.match(r"(\w+)\[(\d+)\]", ct)
            expect = C[arr.group(1)] * int(arr.group(2)) if arr else C[ct]
            assert ctypes.sizeof(gt) == ctypes.sizeof(expect) and gt._type_ == expect._type_, (cname, f)
    # every field of the statistics is 8 bytes wide, or an array of such (the per-kernel-template counts): no padding anywhere
    assert ctypes.sizeof(api.CStats) == sum(ctypes.sizeof(t) for _, t in api.CStats._fields_)
    assert all(ctypes.sizeof(t) % 8 == 0 for _, t in api.CStats._fields_)


def test_settings_defaults_match_reference():
    # libs/dmrecon/settings.h:25-51
    L = api.load_library()
    cs = api.CSettings()
    L.mi_dmrecon_settings_default(ctypes.byref(cs))
    assert (cs.filterWidth, cs.maxIterations, cs.nrReconNeighbors, cs.globalVSMax, cs.scale, cs.useColorScale) == (5, 20, 4, 20, 0, 1)
    assert np.allclose([cs.minNCC, cs.minParallax, cs.acceptNCC, cs.minRefineDiff], [0.3, 10.0, 0.6, 0.001])
    assert cs.aabbMin[0] == -np.finfo(np.float32).max and cs.aabbMax[2] == np.finfo(np.float32).max
    s = api.Settings()
    c2 = s.to_c()
    for f, _ in api.CSettings._fields_:
        a, b = getattr(cs, f), getattr(c2, f)
        if hasattr(a, "__len__"):
            assert list(a) == list(b)
        else:
            assert a == pytest.approx(b)
    assert s.imageEmbedding == "undistorted" and not s.keepDzMap and not s.keepConfidenceMap


def test_no_gpu_means_loud_failure_not_fallback():
    if api.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no HIP device"):
        api.Context(0)
    L = api.load_library()
    h = ctypes.c_void_p()
    assert L.mi_dmrecon_ctx_create(0, ctypes.byref(h)) == api.E_DEVICE
    assert b"no CPU path" in L.mi_dmrecon_last_error()


class _FakeScene:
    n_views = 3

    def level_size(self, v, s):
        if s > 4:
            raise ValueError("level out of range")
        return 64, 48


def test_dmrecon_ctor_checks_mirror_reference():
    # libs/dmrecon/dmrecon.cc:37-46,74-75 throw std::invalid_argument -> ValueError
    with pytest.raises(ValueError, match="Master view index out of bounds"):
        api.DMRecon(_FakeScene(), api.Settings(refViewNr=3))
    with pytest.raises(ValueError, match="Invalid scale factor"):
        api.DMRecon(_FakeScene(), api.Settings(scale=-1))
    with pytest.raises(ValueError, match="Invalid image embedding"):
        api.DMRecon(_FakeScene(), api.Settings(imageEmbedding=""))
    with pytest.raises(ValueError, match="Invalid master view"):
        api.DMRecon(_FakeScene(), api.Settings(scale=7))
    r = api.DMRecon(_FakeScene(), api.Settings(refViewNr=2, scale=1))
    assert r.getRefViewNr() == 2 and (r.width, r.height) == (64, 48)
    assert r.getProgress().status == 0 and r.getProgress().filled == 0      # RECON_IDLE


def test_oracle_is_not_reachable_from_the_product():
    # the product package must never import, load or link anything under oracle/
    pkg = os.path.join(ROOT, "mve_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".cc", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "liboracle" not in txt and "dmrecon_oracle" not in txt and "from oracle" not in txt \
                    and "import oracle" not in txt, "%s references the oracle" % f


def test_every_run_time_switch_is_documented():
    """Every environment variable the library or the shim reads is described in INTEGRATION.md (and nothing there names
    a switch that no longer exists): the switches are part of the drop-in's interface."""
    import re
    src = ""
    for rel in ("mve_amd/csrc/dmrecon_host.cpp", "mve_amd/host/dmrecon.cc"):
        src += open(os.path.join(ROOT, rel)).read()
    read = set(re.findall(r'(?:getenv|env_or)\("(MI_DMRECON_[A-Z_]+)"', src))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    named = set(re.findall(r"MI_DMRECON_[A-Z_]+", doc))
    codes = {n for n in named if re.match(r"MI_DMRECON_(E[A-Z]+|OK)$", n)}          # return codes, not switches
    assert len(read) >= 12
    assert read <= named, sorted(read - named)
    extra = named - read - codes - {"MI_DMRECON_LIB"}                              # MI_DMRECON_LIB: mve_amd/api.py
    assert not extra, sorted(extra)


def test_bench_call_plan():
    """bench.py: how K timed steps become library calls (pure host logic)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for steps in range(1, 130):
        spc, calls, threads = b.plan_calls(steps, 6)
        assert spc * calls == steps and 1 <= spc <= 5 and 1 <= threads <= min(6, calls)
    assert b.plan_calls(60, 6) == (5, 12, 6) and b.plan_calls(5, 6) == (5, 1, 1) and b.plan_calls(7, 6) == (1, 7, 6)
    assert b.plan_calls(12, 3, 1) == (1, 12, 3) and b.plan_calls(60, 6, 4) == (4, 15, 6)
    with pytest.raises(SystemExit):
        b.plan_calls(10, 6, 4)


# ---- the host half of a call's planning, without a GPU ---------------------------------------------------------------

def _host_views(scene, ref, tables, **kw):
    return api.plan_views_host(scene, api.Settings(refViewNr=ref, **kw), ref, tables=tables)[0]


def test_host_view_selection_is_the_references(g1, g1_scene, h1_scene, w1, w1_scene, w2, w2_scene):
    """GlobalViewSelection as a reconstruct call runs it on the host (plan_global_views: from the scene tables -- dense
    score arrays, factor rows shared per scene, eight in-order sums side by side -- and directly, the form for bundles too
    large for tables) against the restatement, which is bit-identical to the reference: the same views for every
    reference view of the fixtures, with 40 global views, with minParallax / globalVSMax / scale / bounding box changed."""
    from oracle import oracle as orc
    for scene, kws in ((g1_scene, [dict(), dict(globalVSMax=2), dict(minParallax=25.0), dict(scale=1)]),
                       (h1_scene, [dict(), dict(minParallax=4.0)]),
                       (w1_scene, [dict(globalVSMax=40), dict(globalVSMax=40, minParallax=3.0), dict()])):
        S = orc.OracleScene(scene)
        n = len(scene.cameras)
        for kw in kws:
            okw = dict(global_max=kw.get("globalVSMax", 20), scale=kw.get("scale", 0))
            if "minParallax" in kw:
                okw["minParallax"] = kw["minParallax"]
            for ref in range(n) if n <= 9 else (0, 7, 20, 41):
                want = S.global_vs(orc.make_settings(ref_view=ref, **okw))
                for tables in (True, False):
                    assert _host_views(scene, ref, tables, **kw) == want, (n, kw, ref, tables)
    assert _host_views(w1_scene, 0, True, globalVSMax=40) == list(w1["gvs40"])          # the reference binary's own list
    # more than 64 global views (scene W2, 100 views; apps/dmrecon -n 80): the reference binary's own list, both paths, and
    # the library's limit itself (MI_DMRECON_MAX_GLOBAL_VIEWS = 128: every other view of the scene)
    S2 = orc.OracleScene(w2_scene)
    for tables in (True, False):
        assert _host_views(w2_scene, 0, tables, globalVSMax=80) == list(w2["gvs80"]) and len(w2["gvs80"]) == 80
    for ref in (37, 99):
        want = S2.global_vs(orc.make_settings(ref_view=ref, global_max=128))
        assert len(want) > 64 and _host_views(w2_scene, ref, True, globalVSMax=128) == want
    # a bounding box that cuts the features (dmrecon.cc:190-193)
    pos = np.array([f.pos for f in g1_scene.features], np.float32)
    lo, hi = np.percentile(pos, 20, axis=0), np.percentile(pos, 85, axis=0)
    S = orc.OracleScene(g1_scene)
    st = orc.make_settings(ref_view=1)
    st.aabbMin[:], st.aabbMax[:] = [float(v) for v in lo], [float(v) for v in hi]
    want = S.global_vs(st)
    for tables in (True, False):
        assert _host_views(g1_scene, 1, tables, aabbMin=list(lo), aabbMax=list(hi)) == want
    with pytest.raises(ValueError):
        _host_views(g1_scene, 7, True)                                                    # master view out of bounds
    # the seeds (the host half of processFeatures): the reference table and the scan of the features' view lists agree
    for scene, kw in ((g1_scene, dict()), (w1_scene, dict(globalVSMax=3)), (h1_scene, dict(scale=1))):
        for ref in (0, len(scene.cameras) - 1):
            a = api.plan_views_host(scene, api.Settings(refViewNr=ref, **kw), ref, tables=True, seeds=True)
            b = api.plan_views_host(scene, api.Settings(refViewNr=ref, **kw), ref, tables=False, seeds=True)
            assert a[0] == b[0] and len(a[2][0]) > 20
            assert np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][1], b[2][1])


def test_view_selection_of_a_bundle_in_several_parts(g1_scene, h1_scene, w1_scene, monkeypatch):
    """A bundle too large for the dense tables of the whole that falls apart into parts which share no feature (several
    scenes resident in one context: the bench's distinct-scenes variant) gets tables per connected component
    (SceneStore::sub).  A reference view's selection and seeds from its component's tables = the direct form on the merged
    bundle = the restatement on the view's own scene (ids shifted): a view of another part shares no feature with it and is
    never a candidate (global_view_selection.cc:62-101, :44-52)."""
    from oracle import oracle as orc
    from mve_amd.synth import merge_scenes
    parts = [g1_scene, h1_scene, w1_scene]
    big = merge_scenes(parts)
    # (the parts are small: the bound is lowered so that the merged bundle counts as too large, each part does not)
    nv, nf = len(big.cameras), len(big.features)
    limit = max(len(p.cameras) ** 2 * len(p.features) for p in parts) + 1
    assert nv * nv * nf > limit
    monkeypatch.setenv("MI_DMRECON_DEBUG_TABLE_LIMIT", str(limit))
    off = 0
    for part, kw in zip(parts, (dict(), dict(minParallax=4.0), dict(globalVSMax=40))):
        S = orc.OracleScene(part)
        n = len(part.cameras)
        for ref in sorted(set([0, n // 2, n - 1])):
            okw = dict(global_max=kw.get("globalVSMax", 20))
            if "minParallax" in kw:
                okw["minParallax"] = kw["minParallax"]
            want = [v + off for v in S.global_vs(orc.make_settings(ref_view=ref, **okw))]
            got_sub = _host_views(big, ref + off, True, **kw)             # tables of the view's component
            got_dir = _host_views(big, ref + off, False, **kw)            # the direct form
            assert got_sub == want and got_dir == want, (off, ref)
            a = api.plan_views_host(big, api.Settings(refViewNr=ref + off, **kw), ref + off, tables=True, seeds=True)
            b = api.plan_views_host(big, api.Settings(refViewNr=ref + off, **kw), ref + off, tables=False, seeds=True)
            assert a[0] == b[0] and np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][1], b[2][1]) and len(a[2][0]) > 20
        off += n


def test_front_teams_block_map():
    """The block map of a front launch with teams (k_front reads job | member | team size per block): every view has
    exactly its team's members, all on blocks with the same b % n_xcd (one XCD: one L2), an XCD is never asked for more
    workgroups than it has compute units (default limit), and the XCDs that hold fewer views give the larger teams -- to
    the views with the most empty pixels."""
    import collections
    for n_cus in (256, 304, 64, 32):
        n_xcd, cus_x = max(1, n_cus // 32), n_cus // max(1, n_cus // 32)
        for n in list(range(1, 70)) + [100, 255, 256, 400]:
            rng = np.random.RandomState(n)
            empty = rng.randint(0, 50000, n)
            ft = api.front_teams(n, n_cus, 32, empty)
            per_xcd = (n + n_xcd - 1) // n_xcd
            if cus_x // per_xcd <= 1:
                assert ft["team_min"] == 1 and ft["team_max"] == 1 and ft["grid"] == 0, (n_cus, n)
                continue
            assert ft["team_min"] == min(32, cus_x // per_xcd) and ft["grid"] == len(ft["map"]) and ft["grid"] % n_xcd == 0
            jobs = collections.defaultdict(list)
            used = collections.Counter()
            for b, blk in enumerate(ft["map"]):
                if blk is not None:
                    jobs[blk[0]].append((blk[1], blk[2], b % n_xcd))
                    used[b % n_xcd] += 1
            assert sorted(jobs) == list(range(n)), (n_cus, n)
            for j, ms in jobs.items():
                T = ms[0][1]
                assert all(t == T for _, t, _ in ms) and sorted(m for m, _, _ in ms) == list(range(T)), (n_cus, n, j)
                assert len({x for _, _, x in ms}) == 1 and ft["team_min"] <= T <= ft["team_max"], (n_cus, n, j)
            assert max(used.values()) <= cus_x, (n_cus, n, used)
            # the more a view has left to fill, the larger (never the smaller) its team
            order = sorted(range(n), key=lambda j: (-empty[j], j))
            sizes = [jobs[j][0][1] for j in order]
            assert sizes == sorted(sizes, reverse=True), (n_cus, n)
    ft = api.front_teams(20, 256, 32, None)
    assert collections.Counter(blk[2] for blk in ft["map"] if blk and blk[1] == 0) == {16: 8, 10: 12}
    ft = api.front_teams(9, 256, 32, None)
    assert collections.Counter(blk[2] for blk in ft["map"] if blk and blk[1] == 0) == {32: 7, 16: 2}
    ft = api.front_teams(5, 256, 8, None)                                   # MI_DMRECON_FRONT_TEAM=8: an upper limit
    assert ft["team_min"] == ft["team_max"] == 8
    assert api.front_teams(20, 256, 1, None)["grid"] == 0


def test_host_view_selection_at_config3_size():
    """The same check at BASELINE config 3's size (20 views, ~2 000 features: rows of ~2 000 floats through the
    eight-row in-order sums, 19 greedy rounds): all 20 reference views, both paths, and a smaller globalVSMax."""
    from oracle import oracle as orc
    from mve_amd.synth import CONFIGS, make_scene
    cfg = CONFIGS["C3"]
    scene = make_scene(cfg["params"])
    S = orc.OracleScene(scene)
    for ref in range(cfg["params"].n_views):
        want = S.global_vs(orc.make_settings(ref_view=ref, scale=cfg["scale"]))
        assert len(want) == 19
        assert _host_views(scene, ref, True, scale=cfg["scale"]) == want, ref
        if ref in (0, 11):
            assert _host_views(scene, ref, False, scale=cfg["scale"]) == want, ref
            w5 = S.global_vs(orc.make_settings(ref_view=ref, scale=cfg["scale"], global_max=5))
            assert _host_views(scene, ref, True, scale=cfg["scale"], globalVSMax=5) == w5, ref
