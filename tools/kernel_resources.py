"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin or file) per kernel."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
cur, rows = None, {}
for line in txt.splitlines():
    if "Name:" in line:
        cur = re.search(r"Name: (\S+)", line).group(1)
        rows[cur] = {}
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPRs Spill", "TotalSGPRs"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
for k, v in rows.items():
    try:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    except OSError:
        name = k
    name = name.replace("mww::", "").split("(")[0]
    print("%-46s vgpr=%-4s agpr=%-3s sgpr=%-4s scratch=%-4s occ=%-2s lds=%-6s vspill=%s" % (
        name, v.get("VGPRs"), v.get("AGPRs"), v.get("TotalSGPRs"), v.get("ScratchSize [bytes/lane]"),
        v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]"), v.get("VGPRs Spill")))
