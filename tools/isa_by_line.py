#!/usr/bin/env python3
"""Instructions of one kernel grouped by the source line they come from (hipcc -gline-tables-only assembly): the view
that found round 4's per-row leftovers in K2 (a canonicalising v_max before every fminf, a compare + select where a
multiplication rule does, per-row tests that a per-thread watch replaces).

    cd rs-sync_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast-honor-pragmas \\
        -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -gline-tables-only --cuda-device-only -S rssync_kernels.hip -o /tmp/k.s
    python tools/isa_by_line.py /tmp/k.s 'lmeds_kernelILi8ELi0ELi80ELb1' [min vector instructions per line, default 8]
"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
min_v = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = re.search(r"^(_Z\w*%s\w*):(.*?)\.amdhsa_kernel" % re.escape(pat), txt, re.S | re.M)
if not m:
    sys.exit("no kernel matching %r" % pat)
print(m.group(1))
cur = None
hist = collections.defaultdict(collections.Counter)
for line in m.group(2).split("\n"):
    t = line.strip()
    loc = re.match(r"\.loc\s+\d+\s+\d+\s+\d+.*; (\S+):(\d+)", t)
    if loc:
        cur = (loc.group(1).split("/")[-1], int(loc.group(2)))
        continue
    if not t or t[0] in ".;" or t.endswith(":") or cur is None:
        continue
    hist[cur][t.split()[0]] += 1
total = collections.Counter()
for c in hist.values():
    total.update(c)
print("by opcode:", total.most_common(30))
for key in sorted(hist):
    c = hist[key]
    v = sum(n for op, n in c.items() if op.startswith("v_"))
    if v >= min_v:
        print("%s:%d  %d vector  %s" % (key[0], key[1], v, c.most_common(8)))
