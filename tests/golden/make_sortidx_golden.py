"""Generate tests/golden/sort_indexes.json from the REFERENCE's own sort_indexes
(als_CP.cxx:835-843), compiled by `make -C oracle ref` into oracle/_ref/sortidx_ref.

Run in the authoring container only (needs /root/reference); the JSON is the committed fixture.
Inputs: seeded vectors of DISTINCT values of lengths 2..8 (the number of modes), the shape of the
relative-perturbation vectors alsCP_PP_partupdate sorts (std::sort leaves the order of equal
values unspecified, so ties are not part of the fixture)."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
rng = np.random.default_rng(20260303)
inputs = []
for n in range(2, 9):
    for _ in range(6):
        v = rng.random(n) * 10.0 ** rng.integers(-6, 3)
        inputs.append([float(x) for x in v])
text = "\n".join(" ".join(repr(x) for x in v) for v in inputs) + "\n"
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "sortidx_ref")], input=text,
                              text=True)
orders = [[int(t) for t in ln.split()] for ln in out.strip().splitlines()]
assert len(orders) == len(inputs)
path = os.path.join(ROOT, "tests", "golden", "sort_indexes.json")
with open(path, "w") as f:
    json.dump({"source": "reference als_CP.cxx:835-843 via oracle/_ref/sortidx_ref",
               "cases": [{"v": v, "order": o} for v, o in zip(inputs, orders)]}, f, indent=1)
print("wrote", path, len(inputs), "cases")
