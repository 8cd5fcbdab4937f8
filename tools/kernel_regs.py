#!/usr/bin/env python3
"""VGPR / SGPR-spill / scratch figures of the chain and table kernels from a -save-temps gfx950 assembly file."""
import re, sys
src = open(sys.argv[1] if len(sys.argv) > 1 else "build/fusion_capi-hip-amdgcn-amd-amdhsa-gfx950.s").read()
for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count: (\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count: (\d+)", src, re.S):
    n = m.group(1)
    if any(k in n for k in ("fuse_chain", "accept_chain", "table_group", "finalize", "verdict")):
        print("%-60s scratch %4s sgpr %3s sgpr_spill %3s vgpr %3s vgpr_spill %s" % (n[:60], m.group(2), m.group(3), m.group(4), m.group(5), m.group(6)))
