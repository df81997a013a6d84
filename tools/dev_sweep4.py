"""Developer script: RENE_READY_MIN x RENE_LEAF_MIN sweep of the traversal-restart kernel (run once per setting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api
out = []
for nm, sc in (("dragon", scenes.dragon_class(1920, 1080)), ("teapot", scenes.teapot_class(1920, 1080))):
    with api.Renderer(sc) as r:
        r.render(0, 4); r.sync(); r.reset(); r.render(0, 16); r.sync(); st = r.stats()
    out.append(f"{nm} {st.rays/st.kernel_ms/1e3:.0f}")
print(os.environ.get("RENE_READY_MIN"), os.environ.get("RENE_LEAF_MIN"), " ".join(out), flush=True)
