"""Developer script: throughput vs frames-per-launch and vs resident blocks per CU (Cornell 1024^2)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from rene_amd import scenes, api
    s = scenes.cornell_box(1024, 1024)
    r = api.Renderer(s)
    r.render(0, 16); r.sync()
    out = []
    for F in (32, 64, 128, 256, 512):
        r.reset(); t = time.time(); r.render(0, F); r.sync(); dt = time.time() - t
        st = r.stats(); out.append(f"F={F}: {st.rays/st.kernel_ms/1e3:.0f}")
    print(os.environ.get("RENE_BLOCKS_PER_CU", "default"), " | ".join(out), flush=True)
else:
    for b in ("1", "2", "3", "4", "5", "6", "8"):
        env = dict(os.environ, RENE_BLOCKS_PER_CU=b)
        subprocess.run([sys.executable, __file__, "child"], env=env)
