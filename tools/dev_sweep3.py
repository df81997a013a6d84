"""Developer script: traversal-restart thresholds on the dragon-class scene."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from rene_amd import scenes, api
    s = scenes.dragon_class(1920, 1080)
    r = api.Renderer(s)
    r.render(0, 4); r.sync()
    for rm in (1, 8, 16, 24, 32, 48):
        out = []
        for lm in (1, 8, 16, 24, 32):
            os.environ["RENE_READY_MIN"] = str(rm); os.environ["RENE_LEAF_MIN"] = str(lm)
            r.reset(); r.render(0, 8); r.sync(); st = r.stats()
            out.append(f"leaf{lm}: {st.rays/st.kernel_ms/1e3:.0f}")
        print(f"ready_min={rm} blocks={os.environ.get('RENE_BLOCKS_PER_CU','def')}", " | ".join(out), flush=True)
else:
    for b in ("4", "8"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, RENE_BLOCKS_PER_CU=b))
