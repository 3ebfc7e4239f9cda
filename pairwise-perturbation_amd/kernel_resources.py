"""Print VGPR/AGPR/occupancy per kernel of csrc/hip_ops.hip (hipcc -Rpass-analysis)."""
import re, subprocess, sys, os
here = os.path.dirname(os.path.abspath(__file__))
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                      os.path.join(here, "csrc/hip_ops.hip"), "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":",1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":",1); cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ppals::", "")
    print(f"{name:55s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} spill {r.get('VGPRs Spill','?'):>2} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}")
