"""Developer script: quantisation of pixels-per-lane (Cornell, F=256)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from rene_amd import scenes, api
    out = []
    for (w, h) in ((1024, 1024), (1280, 1024), (1024, 768), (1024, 1280), (2048, 1024), (2560, 2048)):
        s = scenes.cornell_box(w, h)
        r = api.Renderer(s)
        r.render(0, 16); r.sync(); r.reset()
        r.render(0, 256); r.sync()
        st = r.stats(); out.append(f"{w}x{h}: {st.rays/st.kernel_ms/1e3:.0f}")
        r.close()
    print(os.environ.get("RENE_BLOCKS_PER_CU", "default"), " | ".join(out), flush=True)
else:
    for b in ("3", "4", "5"):
        env = dict(os.environ, RENE_BLOCKS_PER_CU=b)
        subprocess.run([sys.executable, __file__, "child"], env=env)
