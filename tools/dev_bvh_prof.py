import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
which = sys.argv[1] if len(sys.argv) > 1 else "dragon"
sc = scenes.dragon_class(1920, 1080) if which == "dragon" else scenes.teapot_class(1920, 1080)
with api.Renderer(sc) as r:
    r.render(0, 16); r.sync(); st = r.stats()
    print(f"{which} megakernel: {st.rays/st.kernel_ms/1e3:.0f} Mrays/s, {st.kernel_ms/st.frames:.3f} ms/frame, rays {st.rays}")
