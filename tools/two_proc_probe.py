"""Which way of running the drop-in binary changes bits of its maps (there must be none): alone, two processes on one GPU,
without front teams, with small shim batches, with one executor.  GPU box; prints one line per variant."""
import os, subprocess, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mve_amd import scene_io
from mve_amd.synth import CONFIGS, make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "build", "dmrecon_mi")
cfg = CONFIGS["C3"]; n = cfg["params"].n_views; s = cfg["scale"]
scene = make_scene(cfg["params"])
work = tempfile.mkdtemp(prefix="twoproc_")
src = os.path.join(work, "src"); scene_io.write_scene(src, scene)
cmd = [APP, "-s%d" % s, "--keep-conf", "--keep-dz", "--force", "--progress=silent"]

def maps(d):
    return [(scene_io.read_mvei(os.path.join(scene_io.view_dir(d, v), "depth-L%d.mvei" % s)),
             scene_io.read_mvei(os.path.join(scene_io.view_dir(d, v), "conf-L%d.mvei" % s))) for v in range(n)]

def run(tag, n_proc, env):
    dirs = []
    for k in range(n_proc):
        d = os.path.join(work, "%s_%d" % (tag, k)); shutil.copytree(src, d); dirs.append(d)
    e = dict(os.environ, **env)
    ps = [subprocess.Popen(cmd + [d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e) for d in dirs]
    outs = [p.communicate(timeout=600)[0] for p in ps]
    res = []
    for d, p, o in zip(dirs, ps, outs):
        if p.returncode != 0:
            print(tag, "rc", p.returncode, o[-800:]); res.append(None); continue
        res.append((maps(d), o))
    for d in dirs:
        shutil.rmtree(d)
    return res

ref = run("alone", 1, {})[0][0]
for tag, n_proc, env in (("alone_again", 1, {}), ("two_procs", 2, {}), ("two_procs_no_teams", 2, {"MI_DMRECON_FRONT_TEAM": "1"}),
                         ("two_procs_trace", 2, {"MI_DMRECON_TRACE": "1"}),
                         ("small_batches", 1, {"MI_DMRECON_MAX_BATCH": "7"}), ("small_batches_one_executor", 1, {"MI_DMRECON_MAX_BATCH": "7", "MI_DMRECON_EXECUTORS": "1"}),
                         ("small_batches_no_teams", 1, {"MI_DMRECON_MAX_BATCH": "7", "MI_DMRECON_FRONT_TEAM": "1"}),
                         ("one_launch_0", 1, {"MI_DMRECON_ONE_LAUNCH": "0"})):
    for k, r in enumerate(run(tag, n_proc, env)):
        if r is None:
            continue
        m, o = r
        bad = []
        for v in range(n):
            dd = (m[v][0] != ref[v][0]) | (m[v][1] != ref[v][1])
            if dd.any():
                both = (m[v][0] > 0) & (ref[v][0] > 0)
                rel = np.abs(m[v][0] - ref[v][0])[both] / ref[v][0][both]
                ys, xs = np.nonzero(dd[:, :, 0])
                bad.append((v, int(dd.sum()), float(rel.max()), int(((m[v][0] > 0) != (ref[v][0] > 0)).sum()), (int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max()))))
        gave = o.count("gave up")
        print("%-28s proc %d: %d views differ %s  (teams gave up: %d)" % (tag, k, len(bad), bad[:4], gave))
        if "trace" in tag:
            print("\n".join(l for l in o.splitlines() if "gave up" in l or "front launch" in l or "phase C" in l)[:1500])
shutil.rmtree(work)
