"""Regenerates the committed fixtures under tests/golden/.

Nothing here reads the reference at run time: the PCG32si known answers come from an independent
pure-Python restatement of rene-shader/src/rand.rs:4-52 (also quoted in SURVEY.md section 8a A2);
the render fixtures come from the CPU oracle (oracle/rene_oracle.cpp) and pin it against silent
regressions -- they are NOT reference-produced outputs (the reference cannot run here, see
DESIGN.md "Oracle").
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

M32 = 0xFFFFFFFF


class PyPcg:  # rand.rs:4-52, integer arithmetic only
    def __init__(self, seed):
        self.s = seed & M32
        self._step()
        self.s = (self.s + seed) & M32
        self._step()

    def _step(self):
        self.s = (self.s * 747796405 + 2891336453) & M32

    def next_u32(self):
        o = self.s
        self._step()
        w = (((o >> ((o >> 28) + 4)) ^ o) * 277803737) & M32
        return ((w >> 22) ^ w) & M32


def main():
    kat = {}
    for seed in (0, 1, 42, 0xDEADBEEF, 0x52454E45, 0xFFFFFFFF):
        g = PyPcg(seed)
        state = g.s
        vals = [g.next_u32() for _ in range(8)]
        kat[str(seed)] = {"state_after_new": state, "u32": vals,
                          "f32_bits_num": [v >> 8 for v in vals]}  # next_f32 = (v >> 8) * 2^-24
    with open(os.path.join(HERE, "pcg32si_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    from rene_amd import scenes
    from oracle import oracle
    s = scenes.cornell_box(64, 64)
    o = oracle.Oracle(s)
    o.render(0, 4, threads=1)
    np.save(os.path.join(HERE, "cornell_64x64_4spp_layers.npy"),
            np.stack([o.download(l) for l in range(3)]).astype(np.float32))
    st = o.stats().as_dict()
    with open(os.path.join(HERE, "cornell_64x64_4spp_stats.json"), "w") as f:
        json.dump({k: st[k] for k in ("rays_closest", "rays_emitter", "rays_shadow", "paths", "hits", "adds")}, f)
    rays = {"s": [0.0, 1.0, 0.5, 0.25], "t": [0.0, 1.0, 0.5, 0.75], "o": [], "d": []}
    for a, b in zip(rays["s"], rays["t"]):
        ro, rd = o.camera_ray(a, b)
        rays["o"].append([float(x) for x in ro])
        rays["d"].append([float(x) for x in rd])
    with open(os.path.join(HERE, "cornell_camera_rays.json"), "w") as f:
        json.dump(rays, f, indent=1)

    # the volumetric integrator's oracle, pinned the same way: cornell-fog (fog + a dense cloud behind None-material boundaries)
    sv = scenes.cornell_fog(48, 48)
    ov = oracle.Oracle(sv)
    ov.render(0, 4, threads=1)
    np.save(os.path.join(HERE, "cornell_fog_48x48_4spp_layers.npy"), np.stack([ov.download(l) for l in range(3)]).astype(np.float32))
    stv = ov.stats().as_dict()
    with open(os.path.join(HERE, "cornell_fog_48x48_4spp_stats.json"), "w") as f:
        json.dump({k: stv[k] for k in ("rays_closest", "rays_emitter", "rays_shadow", "paths", "hits", "adds")}, f)

    # per-function vectors of every material kind ("per-function vectors for each BxDF / microfacet / Fresnel function
    # from the CPU oracle"): inputs + the oracle's f, pdf, sample_f for 48 well-conditioned configurations per material
    out = {}
    for tag, scene, mats in (("zoo", scenes.material_zoo(32, 32), (1, 2, 3, 4, 5, 6, 7, 8, 10)), ("veach", scenes.veach_mis(32, 32), (1, 2, 3, 4, 6))):
        oo = oracle.Oracle(scene)
        for m in mats:
            rng = np.random.default_rng(9000 + m)
            n = 48
            nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            loc = lambda: (lambda d: d / np.linalg.norm(d, axis=1, keepdims=True))(np.c_[rng.normal(size=(n, 2)), rng.uniform(0.3, 1.0, n) * rng.choice([-1, 1], n)])
            wo_l, wi_l = loc(), loc()
            wo_l[:, 2] = np.abs(wo_l[:, 2])
            wo, wi = np.zeros((n, 3)), np.zeros((n, 3))
            for i in range(n):
                w = nrm[i]
                a = np.array([0, w[2], -w[1]]) if abs(w[0]) <= abs(w[1]) else np.array([-w[2], 0, w[0]])
                u = a / np.linalg.norm(a); v = np.cross(w, u)
                wo[i] = wo_l[i, 0] * u + wo_l[i, 1] * v + wo_l[i, 2] * w
                wi[i] = wi_l[i, 0] * u + wi_l[i, 1] * v + wi_l[i, 2] * w
            uv = rng.uniform(0, 1, (n, 2))
            seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint32)
            nrm, uv, wo, wi = (x.astype(np.float32) for x in (nrm, uv, wo, wi))
            res = np.zeros((n, 12), np.float32)
            for i in range(n):
                e = oo.bsdf_eval(m, nrm[i], uv[i], wo[i], wi[i], int(seeds[i]))
                res[i] = np.concatenate([e["f"], [e["pdf"]], e["s_wi"], e["s_f"], [e["s_pdf"]], [e["len"]]])
            out[f"{tag}_{m}"] = np.concatenate([nrm, uv, wo, wi, seeds.view(np.float32)[:, None], res], axis=1)  # [n][3+2+3+3+1+12]
    np.savez_compressed(os.path.join(HERE, "bxdf_vectors.npz"), **out)


if __name__ == "__main__":
    main()
