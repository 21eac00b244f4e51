"""Which kernels of the SHIPPED library carry scratch (private segment bytes per lane)?
usage: python tools/isa/scratch_audit.py [microwakeword_amd/libmww_hip.so]
Pulls the gfx950 code objects out of the library's clang offload bundles and reads every kernel's
.private_segment_fixed_size from the code-object notes (llvm-readelf).  Where the reloads sit (inside the tile loop = a
vmcnt(0) wait for the prefetch, see tools/isa/summ.py) is answered by tools/isa/dump.sh for the instantiation in question."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    for m in re.finditer(re.escape(MAGIC), data):
        p = m.start()
        o = p + len(MAGIC)
        (count,) = struct.unpack_from("<Q", data, o)
        o += 8
        for _ in range(count):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                yield data[p + off:p + off + size]


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "..", "microwakeword_amd", "libmww_hip.so")
    rows, total = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for i, blob in enumerate(code_objects(lib)):
            f = os.path.join(tmp, "co%d.o" % i)
            open(f, "wb").write(blob)
            notes = subprocess.run([READELF, "--notes", f], capture_output=True, text=True).stdout
            for k in re.split(r"\n\s+- ", notes):
                name = re.search(r"\.name:\s+(\S+)", k)
                size = re.search(r"\.private_segment_fixed_size:\s+(\d+)", k)
                if name and size and name.group(1).startswith("_Z"):
                    total += 1
                    if int(size.group(1)):
                        rows.append((int(size.group(1)), name.group(1)))
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in rows), capture_output=True, text=True).stdout.split("\n")
    print("%s: %d kernels, %d with scratch" % (os.path.basename(lib), total, len(rows)))
    for (size, _), dem in sorted(zip(rows, names)):
        print("%5d  %s" % (size, dem.replace("mww::", "")))


if __name__ == "__main__":
    main()
