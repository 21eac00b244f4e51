// mww_version() of include/mww.h.  Its own translation unit: build_native.py compiles the sha256 of the source set
// (csrc/* + include/mww.h) into it, so that a shipped libmww_hip.so can be checked against the tree it claims to come
// from (bench.py prints both hashes; __graft_entry__.build() recompiles when they differ).
#include "../../include/mww.h"

#ifndef MWW_SOURCE_SHA
#define MWW_SOURCE_SHA "unstamped"
#endif

extern "C" const char* mww_version(void) { return "mww-hip 0.1 (gfx950) src=" MWW_SOURCE_SHA; }
