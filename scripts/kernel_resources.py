"""Registers / LDS / occupancy of every kernel of one .hip file, as the compiler reports them (-Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py csrc/<file>.hip [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/lib/llvm/bin/clang++", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-I" + os.path.join(ROOT, "include"), "-c", "-x", "hip", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only"]
err = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.join(ROOT, "arrowspace-rs_amd")).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r">\(.*", ">", cur).replace("asx::(anonymous namespace)::", "").replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[a-zA-Z/]*\])?: (\S+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    if flt in k:
        print("%-60s vgpr %s agpr %s sgpr %s spill %s/%s scratch %s lds %s occ %s" % (k[:60], v.get("VGPRs"), v.get("AGPRs"), v.get("TotalSGPRs"),
              v.get("VGPRs Spill"), v.get("SGPRs Spill"), v.get("ScratchSize"), v.get("LDS Size"), v.get("Occupancy")))
