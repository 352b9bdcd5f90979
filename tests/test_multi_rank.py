"""CPU test of the N > 1 path: two gloo processes run the bench's barrier / shard / max-over-ranks logic."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_views_partition():
    from mve_amd.dist import shard_views
    views = list(range(20))
    for world in (1, 2, 4, 8):
        parts = [shard_views(views, r, world) for r in range(world)]
        assert sorted(v for p in parts for v in p) == views
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_two_rank_gloo_barrier_and_max(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        from mve_amd.dist import Collective, shard_views
        c = Collective("gloo")
        assert c.world == 2
        mine = shard_views(list(range(5)), c.rank, c.world)
        c.barrier()
        t0 = time.perf_counter()
        time.sleep(0.05 * (c.rank + 1))                 # rank 1 is the slow one
        c.barrier()
        el = c.max(0.05 * (c.rank + 1))
        total = c.sum(len(mine))
        assert abs(el - 0.10) < 1e-9, el
        assert int(total) == 5
        if c.rank == 0:
            print("OK", el, int(total))
        c.close()
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK 0.1" in out.stdout


def test_two_rank_gloo_strong_and_weak_modes(tmp_path):
    """bench.py's two multi-GPU modes on two gloo ranks with a stand-in for the library call (time ~ views):
    weak = every rank all 5 views, strong = the 5 views of one scene dealt 3 + 2; value = maps of all ranks /
    slowest rank, exactly as bench.py computes it."""
    script = tmp_path / "worker2.py"
    script.write_text(textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        from mve_amd.dist import Collective, rank_views
        c = Collective("gloo")
        views, steps, per_view = list(range(5)), 4, 0.01
        out = {}
        for mode in ("weak", "strong"):
            mine = rank_views(views, c.rank, c.world, mode)
            c.barrier()
            t0 = time.perf_counter()
            time.sleep(per_view * len(mine) * steps)       # the reconstruct calls
            c.barrier()
            el = c.max(time.perf_counter() - t0)
            n = int(round(c.sum(len(mine) * steps)))
            out[mode] = (n, n / el, mine)
        assert out["weak"][0] == 2 * 5 * steps and out["strong"][0] == 5 * steps
        assert out["strong"][2] == ([0, 2, 4] if c.rank == 0 else [1, 3])
        # weak: 40 maps in ~0.2 s; strong: 20 maps in ~0.12 s (the rank with 3 views is the slow one)
        assert 150 < out["weak"][1] < 210 and 120 < out["strong"][1] < 175, out
        if c.rank == 0:
            print("MODES_OK")
        c.close()
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29618", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "MODES_OK" in out.stdout
