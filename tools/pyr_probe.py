import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import scene_from_golden
from mve_amd import api
from oracle import oracle as orc
g = dict(np.load("/root/repo/tests/golden/g1b_5views_322x241_scale1.npz"))
sc = scene_from_golden(g)
ctx = api.Context(0); ctx.load_scene(sc)
S = orc.OracleScene(sc)
for lvl in range(ctx.num_levels(2)):
    a = ctx.get_level(2, lvl)[0]; b = S.pyramid_level(2, lvl)[0]
    d = np.argwhere(a != b)
    print(lvl, a.shape, "mismatch", len(d), d[:8].tolist(), [(int(a[tuple(i)]), int(b[tuple(i)])) for i in d[:8]])
