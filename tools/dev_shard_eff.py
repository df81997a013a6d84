"""Developer script: per-GPU efficiency of the tile-sharded bench workload, measured on ONE GPU by
rendering only rank 0's share for shard_count = 1, 2, 4, 8 (no exchange step).  eff = t1 / (N * tN)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi

s = scenes.cornell_box(1024, 1024).to_desc()
F, K = 64, 16
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
mode = abi.SHARD_FRAMES if len(sys.argv) > 2 and sys.argv[2] == "frames" else abi.SHARD_TILES
print("mode", "frames" if mode == abi.SHARD_FRAMES else "tiles", "flags", flags)
base = None
for n in (1, 2, 4, 8):
    with api.Renderer(s, flags=flags, shard_mode=mode, shard_rank=0, shard_count=n) as r:
        r.render(0, F); r.sync(); r.reset()
        t = time.perf_counter()
        for k in range(K):
            r.render(k * F, F)
        r.sync()
        dt = time.perf_counter() - t
        st = r.stats()
    base = base or dt
    print(f"shards {n}: {dt*1e3:.2f} ms wall, kernel {st.kernel_ms:.2f} ms, {st.rays/dt/1e6:.0f} Mrays/s on this GPU, "
          f"efficiency {base/(n*dt):.3f}", flush=True)
