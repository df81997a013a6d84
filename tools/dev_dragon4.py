import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
s = scenes.dragon_class(1920, 1080)
for flags, name in ((0, "two-level"), (abi.FLAG_SINGLE_LEVEL, "single-level")):
    r = api.Renderer(s, flags=flags)
    r.render(0, 4); r.sync()
    for F in (8, 16, 32, 64):
        r.reset(); r.render(0, F); r.sync(); st = r.stats()
        print(name, F, f"{st.kernel_ms:.1f} ms  {st.rays / st.kernel_ms / 1e3:.0f} Mrays/s  {st.kernel_ms/F:.3f} ms/frame", flush=True)
    r.close()
