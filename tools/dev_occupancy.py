"""Developer script: Cornell throughput at the occupancy set by RENE_LDS_PAD (dynamic LDS bytes per block)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api
s = scenes.cornell_box(1024, 1024).to_desc()
with api.Renderer(s) as r:
    r.render(0, 16); r.sync(); r.reset()
    for k in range(4):
        r.render(k * 64, 64)
    r.sync()
    st = r.stats()
    print(f"LDS_PAD={os.environ.get('RENE_LDS_PAD')}: {st.rays/st.kernel_ms/1e3:.0f} Mrays/s, {st.kernel_ms/4:.3f} ms/launch", flush=True)
