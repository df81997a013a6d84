import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
which = sys.argv[1] if len(sys.argv) > 1 else "dragon"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc = scenes.dragon_class(1920, 1080) if which == "dragon" else scenes.teapot_class(1920, 1080)
with api.Renderer(sc, flags=abi.FLAG_WAVEFRONT) as r:
    r.render(0, 4); r.sync(); r.reset()
    r.render(0, F); r.sync(); st = r.stats()
    print(f"{which} wavefront F={F}: {st.rays/st.kernel_ms/1e3:.0f} Mrays/s, {st.kernel_ms/st.frames:.3f} ms/frame", flush=True)
