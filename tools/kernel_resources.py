#!/usr/bin/env python3
"""VGPRs / scratch / LDS of the kernels in hipcc's assembly:  python tools/kernel_resources.py k.s [regex]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "."
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("(anonymous namespace)::", "")
    if not re.search(pat, dn):
        continue
    get = lambda k: re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1)
    print("%-60s vgpr %3s sgpr %3s scratch %5s lds %6s" % (dn[:60], get("next_free_vgpr"), get("next_free_sgpr"),
                                                          get("private_segment_fixed_size"), get("group_segment_fixed_size")))
