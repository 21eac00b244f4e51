"""Builds ``libmww_hip.so`` (and, for the test-suite, the host-side emulator library) from ``csrc/``.

The library is ten translation units (``mww_lib.hip`` + the block-kernel families of ``block_launch.hip.h``, the largest split by width)
compiled in parallel and linked once.  Objects are cached under ``build/obj`` keyed by the sha256 of the flags and of
every file the unit includes, so editing one kernel header recompiles only the units that see it.

The sha256 of the whole source set (``csrc/*`` + ``include/mww.h``) is compiled into the library
(``-DMWW_SOURCE_SHA``; ``mww_version()`` returns it), so a shipped binary can be checked against the tree it claims to
come from: ``source_sha16()`` is the tree's value, ``library_source_sha16(path)`` the binary's.
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "microwakeword_amd", "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(ROOT, "microwakeword_amd", "libmww_hip.so")
OBJDIR = os.path.join(ROOT, "build", "obj")
UNITS = ("version.cpp", "mww_lib.hip", "tu_bwd_first.hip", "tu_bwd64.hip", "tu_bwd48.hip", "tu_bwd32.hip", "tu_fwd.hip", "tu_bwdw.hip", "tu_bwd.hip",
         "sampler.cpp")   # version.cpp first: the one unit that carries the stamp
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-pthread")
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")
EMU_CLANG = "/opt/rocm/lib/llvm/bin/clang++"
EMU_FLAGS = ("-x", "c++", "-std=c++17", "-O2", "-fPIC", "-Wno-unused-value", "-pthread")


def source_files():
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if not f.startswith(".")]
    return files + [os.path.join(INCLUDE, "mww.h")]


def source_sha16() -> str:
    """sha256 (first 16 hex digits) of the source set a library is built from: names and contents, in sorted order."""
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.relpath(path, ROOT).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def library_source_sha16(path: str = LIB):
    """The source sha a built library carries (None: no such file, or a library from before the stamp existed)."""
    if not os.path.isfile(path):
        return None
    with open(path, "rb") as fh:
        m = re.search(rb"mww-hip [0-9.]+ \(gfx950\) src=([0-9a-f]{16})", fh.read())
    return m.group(1).decode() if m else None


def library_sha16(path: str = LIB):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure(path, seen):
    """the unit's own file and every quoted include below it (csrc/, include/, tests/hipemu are the only roots)"""
    path = os.path.normpath(path)
    if path in seen or not os.path.isfile(path):
        return
    seen.add(path)
    with open(path, "r", errors="replace") as fh:
        text = fh.read()
    for inc in _INC.findall(text):
        for base in (os.path.dirname(path), CSRC, INCLUDE):
            cand = os.path.normpath(os.path.join(base, inc))
            if os.path.isfile(cand):
                _closure(cand, seen)
                break


def _unit_key(unit_path, flags, extra_files=()):
    seen = set()
    _closure(unit_path, seen)
    h = hashlib.sha256(" ".join(flags).encode())
    for f in sorted(seen) + sorted(extra_files):
        h.update(f.encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:20]


def _compile_units(compiler, flags, units, tag, extra_key_files=(), jobs=None, verbose=True):
    """compile `units` (absolute paths) to cached objects, at most `jobs` at a time; returns (objects, n compiled)"""
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = jobs or max(1, min(len(units), os.cpu_count() or 4))
    objs, todo = [], []
    for u in units:
        key = _unit_key(u, (compiler,) + tuple(flags), extra_key_files)
        obj = os.path.join(OBJDIR, "%s-%s-%s.o" % (tag, os.path.splitext(os.path.basename(u))[0], key))
        objs.append(obj)
        if not os.path.isfile(obj):
            todo.append((u, obj))
    running, failed = [], []
    t0 = time.time()

    def reap(block):
        for item in list(running):
            p, u, obj, tmp = item
            if block:
                p.wait()
            if p.poll() is None:
                continue
            running.remove(item)
            if p.returncode == 0:
                os.replace(tmp, obj)
            else:
                failed.append(u)
                if os.path.exists(tmp):
                    os.remove(tmp)
            if block:
                return

    for u, obj in todo:
        while len(running) >= jobs:
            reap(False)
            time.sleep(0.2)
        tmp = obj + ".tmp%d" % os.getpid()
        cmd = [compiler] + list(flags) + ["-c", u, "-o", tmp]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        running.append((subprocess.Popen(cmd), u, obj, tmp))
    while running:
        reap(True)
    if failed:
        raise RuntimeError("compilation failed: " + ", ".join(os.path.basename(f) for f in failed))
    if verbose and todo:
        print("[build] %d unit(s) compiled in %.0f s" % (len(todo), time.time() - t0), flush=True)
    # objects of other source states pile up: keep the newest few per unit
    by_unit = {}
    for f in os.listdir(OBJDIR):
        if f.endswith(".o"):
            by_unit.setdefault(f.rsplit("-", 1)[0], []).append(os.path.join(OBJDIR, f))
    for files in by_unit.values():
        files.sort(key=os.path.getmtime, reverse=True)
        for old in files[6:]:
            if old not in objs:
                os.remove(old)
    return objs, len(todo)


def build_library(out: str = LIB, defines=(), slim: bool = False, jobs=None, verbose=True) -> int:
    """hipcc --offload-arch=gfx950 for every unit (cross-compiles without a GPU), then one link.  Returns the number of
    units that had to be compiled (0: every object came from the cache; the link still runs)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = list(HIPCC_FLAGS) + ["-I", INCLUDE, '-DMWW_SOURCE_SHA="%s"' % source_sha16()] + list(defines)
    if slim:
        flags.append("-DMWW_SLIM")
    units = [os.path.join(CSRC, u) for u in UNITS]
    # the stamp changes with any source file, which would recompile every unit on every edit: only version.cpp sees it
    stamp = [f for f in flags if f.startswith("-DMWW_SOURCE_SHA")]
    plain = [f for f in flags if not f.startswith("-DMWW_SOURCE_SHA")]
    objs_a, n_a = _compile_units(hipcc, plain, units[1:], "hip" + ("slim" if slim else ""), jobs=jobs, verbose=verbose)
    objs_b, n_b = _compile_units(hipcc, plain + stamp, units[:1], "hip" + ("slim" if slim else ""), jobs=1, verbose=verbose)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs_b + objs_a + ["-o", out, "-ldl"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return n_a + n_b


def build_emulator(out: str = None, verbose=False, defines=()):
    """TEST INFRASTRUCTURE: the unchanged product sources compiled as host C++ against tests/hipemu (fibers + emulated
    wave ops).  Returns the library path, or None without clang++."""
    out = out or os.path.join(EMU_DIR, "libmww_emu.so")
    if not os.path.isfile(EMU_CLANG):
        return None
    flags = list(EMU_FLAGS) + ["-I", EMU_DIR, "-I", INCLUDE] + list(defines)
    units = [os.path.join(CSRC, u) for u in UNITS] + [os.path.join(EMU_DIR, "hipemu.cpp")]
    extra = [os.path.join(EMU_DIR, "hip", "hip_runtime.h")]
    objs, n = _compile_units(EMU_CLANG, flags, units, "emu", extra_key_files=extra, verbose=verbose)
    if n or not os.path.isfile(out) or any(os.path.getmtime(o) > os.path.getmtime(out) for o in objs):
        subprocess.run([EMU_CLANG, "-shared", "-fPIC", "-pthread"] + objs + ["-o", out, "-ldl"], check=True)
    return out


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--out", default=LIB)
    ap.add_argument("--slim", action="store_true", help="default-topology kernels only (kernel-tuning builds)")
    ap.add_argument("--emulator", action="store_true")
    ap.add_argument("defines", nargs="*", help="extra -D... flags")
    args = ap.parse_args()
    if args.emulator:
        print(build_emulator(args.out if args.out != LIB else None, verbose=True, defines=args.defines))
    else:
        n = build_library(args.out, defines=args.defines, slim=args.slim)
        print("[build] %s: %d unit(s) compiled, library sha256_16 %s, source sha16 %s" % (args.out, n, library_sha16(args.out), library_source_sha16(args.out)))
    sys.exit(0)
