"""Developer script: Cornell 1024^2, 64-frame launches, one at a time, for RENE_LEVELS in argv (timing only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api
s = scenes.cornell_box(1024, 1024)
out = []
for L in sys.argv[2:]:
    os.environ["RENE_LEVELS"] = L
    with api.Renderer(s) as r:
        r.render(0, 4); r.sync(); r.reset()
        for k in range(6):
            r.render(k * 64, 64)
        r.sync()
        st = r.stats()
        out.append(f"L={L}: {st.rays / st.kernel_ms / 1e3:.0f}")
print(sys.argv[1], " ".join(out), flush=True)
