#!/bin/bash
# per-kernel count of flat_* vs global_* instructions
cd /root/repo/wittgenstein_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -Wno-unused-function -Wno-unused-result "${@:2}" -x hip engine.hip -o /tmp/engine.s 2>/dev/null
python3 - "$1" <<'PY'
import re, sys, subprocess
pat = sys.argv[1]
cur = None; rows = {}
for l in open('/tmp/engine.s'):
    m = re.match(r'^(_ZN?[A-Za-z0-9_]+):', l)
    if m: cur = m.group(1); rows[cur] = [0, 0, 0, 0]; continue
    if cur is None: continue
    if l.startswith('.Lfunc_end'): cur = None; continue
    if re.search(r'\bflat_(load|store|atomic)', l): rows[cur][0] += 1
    elif re.search(r'\bglobal_(load|store|atomic)', l): rows[cur][1] += 1
    elif re.search(r'\bscratch_', l): rows[cur][2] += 1
    if re.search(r's_waitcnt vmcnt\([1-9]', l): rows[cur][3] += 1
names = subprocess.run(['c++filt'] + list(rows), capture_output=True, text=True).stdout.splitlines()
print('%-64s %5s %6s %7s %9s' % ('kernel', 'flat', 'global', 'scratch', 'vmcnt(N>0)'))
for (k, v), n in zip(rows.items(), names):
    n = re.sub(r'\(.*', '', n.replace('void ', '').replace('wg::', ''))
    if re.search(pat, n): print('%-64s %5d %6d %7d %9d' % (n[:64], *v))
PY
