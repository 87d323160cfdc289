#!/usr/bin/env python3
"""Print VGPR/SGPR/spill/LDS per kernel from a device assembly file (hipcc -S --cuda-device-only)."""
import re, subprocess, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
meta = s[s.index('amdhsa.kernels'):]
for e in re.split(r'\n  - \.agpr_count', meta)[1:]:
    name = re.search(r'\.name:\s+(\S+)', e).group(1)
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    if pat and pat not in d:
        continue
    g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, e).group(1))
    print("vgpr %3d sgpr %3d spill %d  %s" % (g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), d.replace('void sp::', '')))
