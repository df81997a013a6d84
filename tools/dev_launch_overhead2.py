"""Developer script: launch time vs image size at fixed frames per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for F in (2, 16):
  for res in (64, 128, 256, 512, 1024, 2048):
    with api.Renderer(scenes.cornell_box(res, res).to_desc(), flags=flags) as r:
        r.render(0, 16); r.sync(); r.reset()
        reps = 16
        for k in range(reps):
            r.render(k * F, F)
        r.sync()
        st = r.stats()
        print(f"F={F} res={res}: {st.kernel_ms/reps*1e3:.1f} us/launch, {st.kernel_ms/reps/F/(res*res)*1e9:.3f} ns/pixel-frame", flush=True)
