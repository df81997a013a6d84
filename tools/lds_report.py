#!/usr/bin/env python3
"""GPU box: what every bench configuration's render kernel keeps in LDS and how many workgroups per CU that leaves (RENE_DEBUG lines of the library)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = ("import sys; sys.path.insert(0, %r)\nimport bench\nfrom rene_amd import api\nlab, mk, spp, fpl = bench.configurations()[sys.argv[1]]\nsc = mk()\n"
        "pk = sc if hasattr(sc, 'byref') else sc.to_desc()\nr = api.Renderer(pk)\nr.render(0, min(spp, int(sys.argv[2])))\nr.sync()\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ("cornell", "veach-mis", "dragon-class", "teapot-class", "dragon-partial", "material-zoo"):
    for frames in ("1024", "8192", "128"):
        p = subprocess.run([sys.executable, "-c", code, name, frames], env=dict(os.environ, RENE_DEBUG="1"), capture_output=True, text=True, timeout=300)
        for l in p.stderr.splitlines():
            if "co-resident" in l or "seed tables" in l:
                print(name, frames, l, flush=True)
