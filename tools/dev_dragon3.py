import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rene_amd import scenes, api, abi
s = scenes.dragon_class(1920, 1080)
for name, flags in (("restart", 0), ("while-while", abi.FLAG_NO_RESTART)):
    r = api.Renderer(s, flags=flags)
    r.render(0, 4); r.sync(); r.reset()
    r.render(0, 16); r.sync()
    st = r.stats(); print(name, f"{st.rays / st.kernel_ms / 1e3:.0f} Mrays/s", f"{st.kernel_ms:.1f} ms", flush=True)
    img = r.download(0)
    if name == "restart": ref = img
    else: print("bit-identical images:", np.array_equal(ref, img), "max abs diff", np.abs(ref - img).max())
    r.close()
