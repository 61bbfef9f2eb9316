#!/bin/bash
# Static look at k_mask_annotate_q20<true, true, 15, 1> (the one-sweep form; the batched launch inlines the same body) without a GPU: registers, spills, instruction classes, spill reloads inside the read loop.
#   tools/k2_isa.sh [out.s]      (the kernel's ISA is left in /tmp/isa/q20.s or the given file)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa
OUT=${1:-/tmp/isa/q20.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -Wno-unused-function --cuda-device-only -S -o /tmp/isa/k.s $R/hinge_amd/csrc/hinge_capi.hip 2>/dev/null || exit 1
python3 - "$OUT" <<'PY'
import re, sys
s = open('/tmp/isa/k.s').read().split('\n')
name = None
for i, l in enumerate(s):
    if re.match(r'^_ZN5hinge19k_mask_annotate_q20ILb1ELb1ELi15ELi1EE.*:\s', l):
        name = l.split(':')[0]; a = i
    if name and l.startswith('\t.amdhsa_kernel ' + name): b = i; break
body = s[a:b]
open(sys.argv[1], 'w').write('\n'.join(body))
meta = '\n'.join(s)
m = re.search(r'\.name:\s+' + re.escape(name) + r'\n(?:.*\n){1,12}', meta)
for key in ('sgpr_count', 'sgpr_spill_count', 'vgpr_count', 'vgpr_spill_count'):
    mm = re.search(r'\.%s:\s+(\d+)' % key, m.group(0)); print(key, mm.group(1) if mm else '?', end='  ')
print()
loop = next(i for i, l in enumerate(body) if 'This Loop Header: Depth=1' in l and 'Child' in body[i + 1])
cnt = lambda pat, lines: sum(1 for l in lines if re.match(r'^\s*' + pat, l))
for nm, lines in (('whole', body), ('read loop', body[loop:])):
    print('%-10s valu %4d  salu %4d  ds %3d  readlane %3d  writelane %3d  s_nop %3d  lines %d' % (nm, cnt('v_', lines), cnt('s_', lines), cnt('ds_', lines),
          cnt('v_readlane', lines), cnt('v_writelane', lines), cnt('s_nop', lines), len(lines)))
PY
