"""Developer script: launch time vs frames per launch (Cornell 1024^2), to separate the fixed cost of a
launch (start-up + tail) from the per-frame cost.  argv[1] = flags."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi

s = scenes.cornell_box(1024, 1024).to_desc()
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
print("flags", flags)
with api.Renderer(s, flags=flags) as r:
    r.render(0, 16); r.sync()
    for F in (2, 4, 8, 16, 32, 64, 128):
        r.reset()
        reps = max(4, 256 // F)
        for k in range(reps):
            r.render(k * F, F)
        r.sync()
        st = r.stats()
        print(f"F={F}: {st.kernel_ms/reps:.3f} ms/launch, {st.kernel_ms/reps/F*1e3:.1f} us/frame", flush=True)
