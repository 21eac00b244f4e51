#!/bin/bash
# tools/ubench/lds_patterns under the SQ LDS counters: conflict cycles / LDS-array cycles per access pattern
# (build first: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/lds_patterns.hip -o tools/ubench/lds_patterns).
# usage (repo root): bash tools/gpu_lds_ubench.sh <tag>
TAG=${1:-ldsub}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT -d $OUT/pmc -o p -- $R/tools/ubench/lds_patterns > $OUT/run.log 2>&1
cd $R
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("%-28s %12s %12s %8s %10s  cycles per wave-instruction (array / conflict)" % ("pattern", "IDX_ACTIVE", "CONFLICT", "ratio", "INSTS_LDS"))
for k in acc:
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    act, con, ins = m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_INSTS_LDS", 1)
    print("%-28s %12.4g %12.4g %7.1f%% %10.4g  %.2f / %.2f" % (k, act, con, 100 * con / max(act, 1), ins, act / max(ins, 1), con / max(ins, 1)))
PY
