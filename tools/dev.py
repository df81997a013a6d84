#!/usr/bin/env python3
"""Developer measurements on a GPU box (one script; `gpurun -- python3 tools/dev.py <command> ...`).

  rate NAME... [--frames F] [--launches N] [--no-tune]   wall-clock Mrays/s of bench.py's configurations
        (plus zoo, fog, media-zoo, dragon-fog, veach-bvh, cornell-bvh), launches one after the other
  sweep VAR V1,V2,... NAME [rate options]                           the same once per value of an environment knob
        (RENE_LEVELS, RENE_READY_MIN, RENE_LEAF_MIN, RENE_BLOCKS_PER_CU, ...): fresh process per value
  soak [--flags 0x200] [--launches 6 x jobs]                         C4 / C5 at their full sample counts, one launch per job: no hand-off may time out,
        every job's image equal to the first one's (0x200: RENE_FLAG_FRAME_GROUPS)
  counters NAME...                                                   per-ray node / primitive / hit counters of the counting pass
"""
import argparse
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def scene_table():
    from rene_amd import abi, scenes

    def vol_cornell():
        s = scenes.cornell_box(1024, 1024)
        s.integrator = abi.INTEGRATOR_VOLPATH
        return s
    return {
        "cornell": (lambda: scenes.cornell_box(1024, 1024), 256, 0),
        "veach-mis": (lambda: scenes.veach_mis(1024, 1024), 256, 0),
        "dragon-class": (lambda: scenes.dragon_class(1920, 1080), 256, 0),
        "teapot-class": (lambda: scenes.teapot_class(1920, 1080), 512, 0),
        "zoo": (lambda: scenes.material_zoo(1024, 768), 32, 0),
        "fog": (lambda: scenes.cornell_fog(1024, 1024), 16, 0),
        "media-zoo": (lambda: scenes.media_zoo(1024, 768), 16, 0),
        "cornell-vol": (vol_cornell, 32, 0),
        "dragon-fog": (lambda: scenes.dragon_fog(1920, 1080), 64, 0),
        "dragon-fog-ww": (lambda: scenes.dragon_fog(1920, 1080), 64, abi.FLAG_NO_RESTART),
        "dragon-ww": (lambda: scenes.dragon_class(1920, 1080), 64, abi.FLAG_NO_RESTART),
        "cornell-bvh": (lambda: scenes.cornell_box(1024, 1024), 64, abi.FLAG_FORCE_BVH),
        "veach-bvh": (lambda: scenes.veach_mis(1024, 1024), 64, abi.FLAG_FORCE_BVH),
    }


def rate(names, frames, launches, serial, tune, extra_flags=0):
    from rene_amd import abi, api
    tab = scene_table()
    for nm in names:
        mk, F, fl = tab[nm]
        F = frames or F
        s = mk()
        flags = fl | extra_flags
        with api.Renderer(s, flags=flags) as r:
            if tune:
                r.tune(F)
            r.render(0, min(F, 4))
            r.sync()
            r.reset()
            t0 = time.perf_counter()
            for k in range(launches):
                r.render(k * F, F)
            r.sync()
            wall = time.perf_counter() - t0
            st = r.stats()
        print(f"{nm}: {st.rays / wall / 1e6:.0f} Mrays/s wall ({launches} x {F} frames, "
              f"{wall * 1e3 / launches:.2f} ms/launch wall, {st.kernel_ms / st.launches:.2f} ms events), {wall * 1e3 / st.frames:.4f} ms/frame, "
              f"rays/path {st.rays / st.paths:.2f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("command")
    ap.add_argument("args", nargs="*")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--launches", type=int, default=6)
    ap.add_argument("--serial", action="store_true")
    ap.add_argument("--no-tune", action="store_true")
    ap.add_argument("--flags", type=lambda v: int(v, 0), default=0)
    a = ap.parse_args()
    if a.command == "rate":
        rate(a.args or ["cornell", "veach-mis", "dragon-class", "teapot-class"], a.frames, a.launches, a.serial, not a.no_tune, a.flags)
    elif a.command == "sweep":
        var, values, names = a.args[0], a.args[1].split(","), a.args[2:]
        for v in values:
            env = dict(os.environ, **{var: v})
            cmd = [sys.executable, os.path.abspath(__file__), "rate", *names, "--launches", str(a.launches), "--frames", str(a.frames), "--flags", str(a.flags)]
            cmd += (["--serial"] if a.serial else []) + (["--no-tune"] if a.no_tune else [])
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            print(f"{var}={v}: " + " | ".join(l for l in p.stdout.splitlines() if l) + (p.stderr[-300:] if p.returncode else ""), flush=True)
    elif a.command == "soak":
        import numpy as np
        from rene_amd import api, scenes
        for nm, s, spp in (("dragon-class 1920x1080 @ 1024 spp", scenes.dragon_class(1920, 1080), 1024),
                           ("teapot-full 1920x1080 @ 8192 spp", scenes.teapot_full(1920, 1080), 8192)):
            with api.Renderer(s, flags=a.flags) as r:  # (--flags 0x200: RENE_FLAG_FRAME_GROUPS, two chains of frames per pixel)
                first = None
                for j in range(max(1, a.launches // 6)):  # --launches 6 (the default): one job; 60: ten
                    r.reset()
                    t0 = time.perf_counter()
                    r.render(0, spp)  # one launch per job
                    r.sync()
                    dt = time.perf_counter() - t0
                    st = r.stats()  # raises if a hand-off timed out and its replay did not repair it
                    img = r.download(0)
                    first = img if first is None else first
                    print(f"{nm}: job {j}: {dt:.2f} s, {st.rays / dt / 1e6:.0f} Mrays/s, {dt * 1e3 / spp:.3f} ms/frame, launches {st.launches} (1 = none replayed), "
                          f"finite {bool(np.isfinite(img).all())}, mean {img.mean() / spp:.4f}, equal to the first job's image: {bool(np.array_equal(img, first))}", flush=True)
    elif a.command == "counters":
        from rene_amd import abi, api
        tab = scene_table()
        for nm in a.args:
            mk, F, fl = tab[nm]
            with api.Renderer(mk(), flags=fl | abi.FLAG_COUNTERS) as rc:
                rc.render(0, 2)
                c = rc.stats().as_dict()
            print(nm, {k: round(c[k] / c["rays"], 3) for k in ("node_visits", "prim_tests", "hits", "adds", "rays_closest", "rays_shadow", "rays_emitter")},
                  "B_alg/ray", round(abi.algorithmic_bytes(c) / c["rays"], 1), flush=True)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
