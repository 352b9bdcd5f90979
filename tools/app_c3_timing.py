"""Wall time of the drop-in binary (build/dmrecon_mi = MVE's unmodified apps/dmrecon on the shim) on the C3 scene
written as an MVE directory; prints the app's own 'Reconstruction took' line."""
import os, subprocess, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mve_amd import scene_io
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS["C3"]
sc = make_scene(cfg["params"])
work = tempfile.mkdtemp(prefix="c3app_")
sdir = os.path.join(work, "scene")
t0 = time.time(); scene_io.write_scene(sdir, sc); print("scene written in %.1f s" % (time.time() - t0))
app = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "dmrecon_mi")
for rep in range(2):
    t0 = time.time()
    out = subprocess.run([app, "-s2", "--force", "--progress=silent", sdir], capture_output=True, text=True)
    took = [l for l in out.stdout.splitlines() if "Reconstruction took" in l]
    print("run %d: wall %.2f s rc %d; %s" % (rep, time.time() - t0, out.returncode, took[-1] if took else out.stdout[-300:] + out.stderr[-300:]))
    if os.environ.get("MI_DMRECON_TRACE"):                  # the shim's host-side phases (mve_amd/host/dmrecon.cc)
        lines = [l for l in out.stderr.splitlines() if "shim" in l]
        keep = [l for l in lines if "(view)" not in l] + [l for l in lines if "(view)" in l][:6] + [l for l in lines if "(view)" in l][-6:]
        print("\n".join(sorted(set(keep), key=lambda l: float(l.split("t=")[1].split("ms")[0]))))
shutil.rmtree(work)
