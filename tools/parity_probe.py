"""Where do the HIP path and the reference disagree on WHICH pixels get a depth?  (C3, all views.)
Runs oracle/_ref/dmrecon_ref_fast on the scene, the HIP path on the same scene, and prints per view the fill IoU and,
for the worst views, what the one-sided pixels look like (confidence, position, neighbourhood)."""
import os, sys, tempfile, subprocess, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import api
from mve_amd.scene_io import read_mvei, view_dir, write_scene
from mve_amd.synth import CONFIGS, make_scene, true_depth

cfg = CONFIGS["C3"]; p = cfg["params"]; s = cfg["scale"]
scene = make_scene(p)
ctx = api.Context(0); ctx.load_scene(scene)
st = api.Settings(scale=s)
res = ctx.reconstruct(st, list(range(p.n_views)))
work = tempfile.mkdtemp(prefix="pp_"); sdir = os.path.join(work, "scene"); write_scene(sdir, scene)
flav = sys.argv[1] if len(sys.argv) > 1 else "fast"
subprocess.run([os.path.join("oracle", "_ref", "dmrecon_ref_" + flav), "-s%d" % s, "--force", "--progress=silent", "--keep-conf", sdir],
               check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS="32"))
rows = []
for v in range(p.n_views):
    rd = read_mvei(os.path.join(view_dir(sdir, v), "depth-L%d.mvei" % s))[:, :, 0]
    rc = read_mvei(os.path.join(view_dir(sdir, v), "conf-L%d.mvei" % s))[:, :, 0]
    gd, gc = res[v]["depth"], res[v]["conf"]
    a, b = gd > 0, rd > 0
    iou = (a & b).sum() / max((a | b).sum(), 1)
    rows.append((iou, v, int((a & ~b).sum()), int((~a & b).sum()), int(a.sum()), int(b.sum()), gd, gc, rd, rc))
print("view iou gpu_only ref_only gpu_filled ref_filled")
for r in rows:
    print("%2d %.4f %6d %6d %6d %6d" % (r[1], r[0], r[2], r[3], r[4], r[5]))
rows.sort(key=lambda r: r[0])
for iou, v, go, ro, na, nb, gd, gc, rd, rc in rows[:3]:
    a, b = gd > 0, rd > 0
    g_only, r_only = a & ~b, ~a & b
    gt = true_depth(p, scene.cameras[v], gd.shape[1], gd.shape[0])
    def desc(m, d, c):
        if not m.any():
            return "none"
        ys, xs = np.nonzero(m)
        err = np.abs(d[m] - gt[m])
        return ("n %d, conf mean %.3f min %.3f, |depth - truth| median %.4f, rows %d..%d cols %d..%d"
                % (m.sum(), c[m].mean(), c[m].min(), np.median(err), ys.min(), ys.max(), xs.min(), xs.max()))
    print("view %d iou %.4f" % (v, iou))
    print("  gpu-only:", desc(g_only, gd, gc))
    print("  ref-only:", desc(r_only, rd, rc))
    both = a & b
    print("  both: conf gpu %.4f ref %.4f ; err gpu %.4f ref %.4f" % (gc[both].mean(), rc[both].mean(), np.median(np.abs(gd[both] - gt[both])), np.median(np.abs(rd[both] - gt[both]))))
    # coarse map of one-sided pixels (16 x 9 cells)
    H, W = gd.shape
    for name, m in (("gpu-only", g_only), ("ref-only", r_only)):
        cells = m[: H // 9 * 9, : W // 16 * 16].reshape(9, H // 9, 16, W // 16).sum((1, 3))
        print("  %s per cell:" % name); print("\n".join("   " + " ".join("%4d" % c for c in row) for row in cells))
shutil.rmtree(work)
