"""Instruction-class histogram of every kernel's tile loop (the outermost loop that contains an s_barrier) and of the whole
kernel: how much of the vector-ALU issue stream is arithmetic (FMA / MFMA) and how much is address arithmetic, moves,
selects.  usage: python tools/isa/hist.py <file.s> [name-filter]"""
import collections
import re
import subprocess
import sys

CLASSES = (
    ("mfma", r"v_mfma"),
    ("fma", r"v_(fma|fmac|pk_fma|mac|mad)_f"),
    ("fp_other", r"v_(add|sub|mul|max|min|cvt|exp|log|rcp|rsq|sqrt|med3|pk_add|pk_mul|cmp\w*|cmpx\w*)_(f|pk_f|bf)|v_cvt_"),
    ("mov", r"v_(mov|accvgpr|swap|perm|readfirstlane|readlane|writelane|bfi|alignbit|permlane)"),
    ("select", r"v_cndmask"),
    ("int", r"v_"),            # everything vector that is left: integer / address / bit arithmetic, integer compares
    ("ds_read", r"ds_read|ds_load"),
    ("ds_write", r"ds_write|ds_store"),
    ("vmem", r"buffer_|global_|flat_"),
    ("scratch", r"scratch_"),
    ("barrier", r"s_barrier"),
    ("wait", r"s_waitcnt|s_nop|s_sleep"),
    ("salu", r"s_"),
)


def classify(op):
    for name, pat in CLASSES:
        if re.match(pat, op):
            return name
    return "other"


def hist(lines):
    c = collections.Counter()
    for ln in lines:
        s = ln.strip()
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        c[classify(s.split()[0])] += 1
    return c


def show(tag, c):
    valu = sum(c[k] for k in ("mfma", "fma", "fp_other", "mov", "select", "int"))
    arith = c["mfma"] + c["fma"]
    print("   %-6s valu %4d: mfma %3d fma %3d fp %3d int %3d mov %3d sel %3d | arithmetic %4.1f %% | ds r/w %3d/%3d vmem %3d scratch %2d salu %3d wait %3d barrier %2d"
          % (tag, valu, c["mfma"], c["fma"], c["fp_other"], c["int"], c["mov"], c["select"], 100.0 * arith / max(valu, 1), c["ds_read"],
             c["ds_write"], c["vmem"], c["scratch"], c["salu"], c["wait"], c["barrier"]))


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"\n(_ZN3mww\S+):[^\n]*\n(.*?)\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, body, meta = m.group(1), m.group(2), m.group(3)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("mww::", "")
        if flt not in dem:
            continue
        g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", meta) or [None, "?"])[1]
        lines = body.split("\n")
        # the tile loop: the depth-1 loop that contains an s_barrier - every block the compiler annotates with its header
        start = end = None
        for h in [i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l]:
            mm = re.match(r"\.L(BB\d+_\d+):", lines[h].strip())
            if not mm:
                continue
            tag = "Header=" + mm.group(1) + " "
            marked = [i for i, l in enumerate(lines) if tag in l + " "]
            lo, hi = min(marked + [h]), max(marked + [h])
            while hi + 1 < len(lines) and not re.match(r"\.LBB\d+_\d+:", lines[hi + 1].strip()):
                hi += 1   # to the end of the last block of the loop
            if any("s_barrier" in l for l in lines[lo:hi + 1]):
                start, end = lo, hi
                break
        print("== %s | vgpr %s lds %s scratch %s" % (dem, g("next_free_vgpr"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
        show("kernel", hist(lines))
        if start is not None:
            show("loop", hist(lines[start:end + 1]))


if __name__ == "__main__":
    main()
