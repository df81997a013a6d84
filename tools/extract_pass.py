#!/usr/bin/env python3
"""tools/extract_pass.py DIR -- reduce one rocprofv3 pass (DIR/p_results.db, several MB) to DIR.json: the top_kernels
rows and, for this library's kernels, the per-dispatch sums of every counter.  tools/prof.sh runs it on the GPU box and
deletes the database, so that the whole evidence fits what gpurun copies back."""
import json
import os
import sqlite3
import sys

d = sys.argv[1].rstrip("/")
db = sqlite3.connect(os.path.join(d, "p_results.db"))
out = {"top_kernels": [], "dispatches": []}
try:
    out["top_kernels"] = [list(r) for r in db.execute("select * from top_kernels")]
except Exception as e:  # no kernel ran
    out["error"] = str(e)
try:
    rows = db.execute("select kernel_name, dispatch_id, counter_name, sum(value), max(duration), max(vgpr_count), max(sgpr_count), max(lds_block_size), max(grid_size) "
                      "from counters_collection group by kernel_name, dispatch_id, counter_name order by dispatch_id")
    out["dispatches"] = [list(r) for r in rows if "rene::" in r[0]]
except Exception:
    pass
json.dump(out, open(d + ".json", "w"))
print(f"{d}.json: {len(out['top_kernels'])} kernels, {len(out['dispatches'])} counter rows")
