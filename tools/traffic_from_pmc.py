#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py's output of FETCH_SIZE / WRITE_SIZE passes over one step
of the headline bench): measured HBM bytes per unit of the three kernel classes bench.py prices, keyed by the hash of the
kernel sources they were measured on.  (2 * FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of MI355X_MICROARCH.md.
usage: python tools/traffic_from_pmc.py profiles/r04/pmc_summary.txt <units per launch> > profiles/traffic.json"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

src, units = sys.argv[1], float(sys.argv[2])
text = open(src).read()
blocks = re.split(r"\n(?=\S)", text)
classes = {"extract": "extract1_part_kernel", "scatter": ("subpart32_kernel", "radix_onesweep_kernel<Key1, false, false, true, 22>"),
           "reduce": ("seg_hash_reduce32b_kernel", "seg_hash_reduce32_kernel", "seg_hash_reduce_kernel")}
out = {"kernels_sha": bench.kernels_hash()}
for cls, names in classes.items():
    names = (names,) if isinstance(names, str) else names
    for b in blocks:
        head = b.split("\n", 1)[0]
        if any(head.startswith(n) for n in names):
            f = re.search(r"FETCH_SIZE\s+([0-9.e+]+)", b)
            w = re.search(r"WRITE_SIZE\s+([0-9.e+]+)", b)
            calls = int(re.search(r"calls=(\d+)", head).group(1))
            if not f or not w:
                continue
            kib = (2 * float(f.group(1)) + float(w.group(1))) / calls
            out[cls] = {"bytes_per_unit": round(kib * 1024 / units, 2),
                        "source": "%s: %s, %d launch(es): (2*%s + %s) KiB / %d = %.1f GB per launch over %.4g units"
                                  % (src, head.split("calls")[0].strip(), calls, f.group(1), w.group(1), calls, kib * 1024 / 1e9, units)}
            break
print(json.dumps(out, indent=2))
