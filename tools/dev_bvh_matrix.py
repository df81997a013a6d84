"""Developer script: traversal-restart vs while-while megakernel on every kernel family (BVH forced)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
cases = [("dragon-class 1920x1080 (Matte)", scenes.dragon_class(1920, 1080), 16),
         ("teapot-class 1920x1080 (GEN1)", scenes.teapot_class(1920, 1080), 16),
         ("veach-mis 1024^2 forced BVH (GEN1)", scenes.veach_mis(1024, 1024), 16),
         ("zoo 1024x768 forced BVH (multi-lobe)", scenes.material_zoo(1024, 768), 16),
         ("cornell 1024^2 forced BVH (Matte)", scenes.cornell_box(1024, 1024), 16)]
for name, sc, F in cases:
    out = []
    for tag, flags in (("restart", 0), ("while-while", abi.FLAG_NO_RESTART)):
        with api.Renderer(sc, flags=abi.FLAG_FORCE_BVH | flags) as r:
            r.render(0, 4); r.sync(); r.reset(); r.render(0, F); r.sync(); st = r.stats()
        out.append(f"{tag} {st.rays/st.kernel_ms/1e3:.0f}")
    print(f"{name}: " + ", ".join(out) + " Mrays/s", flush=True)
