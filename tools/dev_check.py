"""Developer smoke script: GPU vs oracle on Cornell + a first timing. Run on a GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rene_amd import scenes, api, abi
from oracle import oracle

def main():
    s = scenes.cornell_box(256, 256)
    o = oracle.Oracle(s)
    r = api.Renderer(s, flags=abi.FLAG_COUNTERS)
    # trace parity
    rng = np.random.default_rng(1)
    n = 20000
    org = np.tile(np.array([[0, 1, 6.8]], np.float32), (n, 1))
    d = np.stack([rng.uniform(-.18, .18, n), rng.uniform(-.18, .18, n), -np.ones(n)], 1).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hg = r.trace(org, d); ho = o.trace(org, d); hb = o.trace(org, d, bruteforce=True)
    print("trace: inst mismatch gpu/oracle", (hg["instance"] != ho["instance"]).sum(), "prim", (hg["primitive"] != ho["primitive"]).sum(),
          "max |dt|", np.abs(hg["t"] - ho["t"]).max(), " oracle bvh vs brute prim mismatch", (hb["primitive"] != ho["primitive"]).sum())
    t = time.time(); r.render(0, 16); r.sync(); print("gpu 256x256x16:", time.time() - t, "s")
    o.render(0, 16)
    a = r.download(0); b = o.download(0)
    sg, so = r.stats().as_dict(), o.stats().as_dict()
    print("gpu", sg); print("ora", so)
    diff = np.abs(a - b); rel = diff / (1 + np.abs(b))
    print("max abs", diff.max(), "pixels >1e-2(1+ref):", (rel.max(axis=2) > 1e-2).sum(), "of", a.shape[0] * a.shape[1],
          "relMSE", float(((a - b) ** 2).sum() / (b ** 2).sum()), "exact-equal px", (diff.max(axis=2) == 0).sum())
    for layer in (1, 2):
        da = np.abs(r.download(layer) - o.download(layer)).max(); print("layer", layer, "max abs diff", da)
    # timing at 1024^2
    s2 = scenes.cornell_box(1024, 1024)
    r2 = api.Renderer(s2)
    r2.render(0, 8); r2.sync()
    for nf in (16, 64):
        r2.reset(); t = time.time(); r2.render(0, nf); r2.sync(); dt = time.time() - t
        st = r2.stats()
        print(f"1024x1024 x{nf}: {dt*1e3:.1f} ms wall, kernel {st.kernel_ms:.1f} ms, rays {st.rays}, {st.rays/st.kernel_ms/1e3:.1f} Mrays/s")

main()
