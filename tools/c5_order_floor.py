import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mve_amd.synth import CONFIGS, make_scene
cfg = CONFIGS["C5"]; p = cfg["params"]
t0 = time.time(); sc = make_scene(p); print("render", time.time() - t0, flush=True)
views = (0, 50)
pid = os.fork()
if pid == 0:
    os.environ["ORC_QUEUE_ORDER"] = "reverse"          # read when the oracle library is loaded (below)
    tag = "rev"
else:
    tag = "fwd"
from oracle import oracle as orc
S = orc.OracleScene(sc)
for v in views:
    t0 = time.time()
    a = S.reconstruct(orc.make_settings(ref_view=v, scale=cfg["scale"], local_neighbors=cfg["local_neighbors"]))
    np.savez("/tmp/c5_%s_%d.npz" % (tag, v), d=a["depth"], c=a["conf"])
    print(tag, v, time.time() - t0, flush=True)
if pid == 0:
    os._exit(0)
os.waitpid(pid, 0)
from conftest import map_parity
out = {}
for v in views:
    a = np.load("/tmp/c5_fwd_%d.npz" % v); b = np.load("/tmp/c5_rev_%d.npz" % v)
    out[str(v)] = {k: float(x) for k, x in map_parity(b["d"], b["c"], a["d"], a["c"]).items()}
    print(v, out[str(v)], flush=True)
json.dump(out, open("/tmp/c5_floor.json", "w"))
