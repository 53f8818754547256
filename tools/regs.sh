#!/bin/bash
# Register / scratch / occupancy table of every kernel of engine.hip, from the compiler's resource-usage remarks
# (cross-compiles for gfx950: no GPU needed). Usage: tools/regs.sh [pattern] [extra hipcc flags...]
cd "$(dirname "$0")/../wittgenstein_amd/csrc"
PAT=${1:-.}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage "$@" -x hip engine.hip -o /tmp/engine_regs.o 2>&1 |
python3 -c '
import re, sys, subprocess
cur = None; rows = []
for l in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", l)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k.split()[0]] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%-70s %5s %5s %7s %4s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("void ", "").replace("wg::", ""))
    if re.search(sys.argv[1], n):
        print("%-70s %5s %5s %7s %4s %6s" % (n[:70], r.get("VGPRs"), r.get("SGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
' "$PAT"
