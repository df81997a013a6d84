import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api
s = scenes.dragon_class(1920, 1080)
r = api.Renderer(s)
r.render(0, 4); r.sync(); r.reset()
r.render(0, 16); r.sync()
st = r.stats(); print(st.rays / st.kernel_ms / 1e3, "Mrays/s", st.kernel_ms, "ms", st.rays, "rays")
