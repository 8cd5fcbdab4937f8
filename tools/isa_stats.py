#!/usr/bin/env python3
"""Instruction mix of one kernel from the -save-temps gfx950 assembly (build/*.s)."""
import collections, re, sys
name = sys.argv[1] if len(sys.argv) > 1 else "observe_frame_kernel"
src = open("build/fusion_capi-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
start = next(i for i, l in enumerate(src) if re.match(r"^_ZN3sfm\d+%s\w*:" % name, l))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
ins = [l.strip() for l in src[start + 1:end] if l.strip() and l.strip()[0] not in ";." and not l.strip().endswith(":")]
c = collections.Counter(l.split()[0] for l in ins)
print(name, "instructions", len(ins))
groups = collections.Counter()
for k, v in c.items():
    if k.startswith("v_") and "f64" in k: groups["valu_f64"] += v
    elif k.startswith("v_"): groups["valu_other"] += v
    elif k.startswith("s_load") : groups["s_load"] += v
    elif k.startswith("s_waitcnt"): groups["s_waitcnt"] += v
    elif k.startswith("s_"): groups["salu"] += v
    elif k.startswith("global_") or k.startswith("flat_") or k.startswith("buffer_"): groups["vmem"] += v
    elif k.startswith("ds_"): groups["lds"] += v
    else: groups[k] += v
print(dict(groups))
for k in ["v_rcp_f64_e32", "v_div_scale_f64", "v_div_fixup_f64", "global_load_ushort", "global_load_dwordx2", "global_load_dwordx4", "s_barrier"]:
    print(" ", k, c.get(k, 0))
for l in src[end:end + 12]:
    if "num_vgpr" in l or "numbered_sgpr" in l or "private_seg" in l: print(l.strip())
