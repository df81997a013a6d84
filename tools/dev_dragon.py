"""Developer script: dragon-class timing at BASELINE config 4 size on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
s = scenes.dragon_class(1920, 1080)
print(api.pack_info(s).as_dict())
t = time.time(); r = api.Renderer(s); print("create", time.time() - t)
r.render(0, 2); r.sync()
for nf in (8, 32):
    r.reset(); t = time.time(); r.render(0, nf); r.sync(); dt = time.time() - t
    st = r.stats(); print(f"1920x1080 x{nf}: {dt*1e3:.1f} ms, {st.rays/dt/1e6:.1f} Mrays/s, rays/path {st.rays/st.paths:.2f}")
with api.Renderer(s, flags=abi.FLAG_COUNTERS) as rc:
    rc.render(0, 4); st = rc.stats()
    print("nodes/ray", st.node_visits / st.rays, "prims/ray", st.prim_tests / st.rays, "B_alg/ray", abi.algorithmic_bytes(st) / st.rays)
