"""Generate tests/golden/dimension_tree.json from the REFERENCE's own Construct_Dimension_Tree
(common.cxx:225-270), compiled by `make -C oracle ref` into oracle/_ref/dimtree_ref.

Run in the authoring container only (needs /root/reference); the JSON is the committed fixture.
"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "dimtree_ref")], text=True)
table = {}
for line in out.strip().splitlines():
    n, recs = line.split(" ", 1)
    nodes = {}
    for rec in recs.strip().strip(";").split(";"):
        key, parent, sibling = rec.split(":")
        nodes[key] = {"parent": parent, "sibling": sibling}
    table[n] = nodes
path = os.path.join(ROOT, "tests", "golden", "dimension_tree.json")
with open(path, "w") as f:
    json.dump({"source": "reference common.cxx:225-270 via oracle/_ref/dimtree_ref", "trees": table},
              f, indent=1, sort_keys=True)
print("wrote", path)
