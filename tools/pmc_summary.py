"""Summarise rocprofv3 CSV output: kernel stats and per-kernel PMC averages.
usage: python tools/pmc_summary.py <dir-with-*_kernel_stats.csv or *_counter_collection.csv> ..."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("mww::", "")
    return name


def library_stamp():
    """First line of every summary: the library the profile was measured on - the sha256 of the file and the sha256 of the
    source set it was built from (mww_version()).  bench.py reports roofline.traffic only from a summary whose library is the
    one it loaded, or was built from the same source set."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "microwakeword_amd", "libmww_hip.so")
    try:
        with open(lib, "rb") as fh:
            data = fh.read()
        m = re.search(rb"mww-hip [0-9.]+ \(gfx950\) src=([0-9a-f]{16})", data)
        return "# library sha256_16=%s source_sha16=%s" % (hashlib.sha256(data).hexdigest()[:16], m.group(1).decode() if m else "unknown")
    except OSError:
        return "# library sha256_16=unknown"


def main():
    print(library_stamp())
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "*kernel_stats.csv")):
            print("## kernel stats", f)
            for row in csv.DictReader(open(f)):
                print("%-46s calls=%-5s avg_us=%9.2f total_us=%10.1f pct=%s" % (
                    short(row["Name"]), row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e3, row["Percentage"]))
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            print("## counters", f)
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            dur = collections.defaultdict(list)
            seen = set()
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                key = (row["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
            names = sorted({c for k in acc for c in acc[k]})
            for k in sorted(acc, key=lambda k: -sum(dur[k])):
                if k.startswith("__amd"):
                    continue
                vals = " ".join("%s=%.4g" % (c, sum(acc[k][c]) / len(acc[k][c])) for c in names if c in acc[k])
                print("%-40s n=%-3d us=%8.2f %s" % (k, len(dur[k]), sum(dur[k]) / len(dur[k]), vals))


if __name__ == "__main__":
    main()
