import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
sc = scenes.teapot_class(1920, 1080).to_desc()
with api.Renderer(sc) as r:
    r.render(0, 4); r.sync(); r.reset()
    worst = 0
    for k in range(0, 64):
        r.render(k, 1); r.sync()
        worst = max(worst, r.stats().last_launch_ms)
    print(f"teapot single-frame launches 0..63: worst {worst:.2f} ms, mean {r.stats().kernel_ms/64:.2f} ms")
