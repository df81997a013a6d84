"""Developer script: throughput + per-ray traversal counters of the two BVH scene classes for whichever
librene_hip.so is in place (used to compare compile-time variants copied over it, one process per variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
tag = sys.argv[1] if len(sys.argv) > 1 else ""
out = []
for nm, sc in (("dragon", scenes.dragon_class(1920, 1080)), ("teapot", scenes.teapot_class(1920, 1080))):
    with api.Renderer(sc) as r:
        r.render(0, 4); r.sync(); r.reset(); r.render(0, 16); r.sync(); st = r.stats()
    with api.Renderer(sc, flags=abi.FLAG_COUNTERS) as rc:
        rc.render(0, 2); c = rc.stats().as_dict()
    out.append(f"{nm} {st.rays/st.kernel_ms/1e3:.0f} Mrays/s nodes/ray {c['node_visits']/c['rays']:.2f} prims/ray {c['prim_tests']/c['rays']:.2f}")
print(tag, " | ".join(out), flush=True)
