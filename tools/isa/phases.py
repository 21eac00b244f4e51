#!/usr/bin/env python3
"""Per-barrier-segment instruction mix of one kernel's main loop in tools/isa/dump.sh's hot.s.
usage: phases.py hot.s <kernel-name-substring>"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
seg, segs = {}, []
def flush(tag):
    global seg
    segs.append((tag, seg)); seg = {}
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"\.LBB\d+_\d+:", t): flush("label " + t)
        continue
    op = t.split()[0]
    if op.startswith("s_barrier"): flush("barrier"); continue
    if op.startswith("v_mfma"): k = "mfma"
    elif op.startswith("ds_read") or op.startswith("ds_load"): k = "ds_read"
    elif op.startswith("ds_write") or op.startswith("ds_store"): k = "ds_write"
    elif op.startswith("buffer_load") or op.startswith("global_load"): k = "vload"
    elif op.startswith("buffer_store") or op.startswith("global_store"): k = "vstore"
    elif op.startswith("v_accvgpr"): k = "accmov"
    elif op.startswith("v_"): k = "valu"
    elif op.startswith("s_waitcnt"): k = "wait"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): k = "branch"
    elif op.startswith("s_"): k = "salu"
    else: k = op
    seg[k] = seg.get(k, 0) + 1
flush("end")
for tag, s in segs:
    if sum(s.values()) >= 8:
        print("%-22s" % tag[:22], " ".join("%s=%d" % kv for kv in sorted(s.items())))
