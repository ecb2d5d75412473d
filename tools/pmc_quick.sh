#!/bin/bash
# two SQ counter passes over a short bench run: bash tools/pmc_quick.sh <tag> [bench args...]
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$ROOT/gpurun_out/pmcq_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3 --latency-batches 2 $*"
pmc() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.log 2>&1; echo $name done; }
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD
pmc sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
src = sys.argv[1]
acc = collections.defaultdict(list)
for fn in glob.glob(f"{src}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "hoqp_solve" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v[-20:]) / len(v[-20:]) for k, v in acc.items()}
wc = c.get("SQ_WAVE_CYCLES", 1)
print({k: "%.4g" % v for k, v in sorted(c.items())})
print("per wave-cycle: active %.3f wait_any %.3f wait_inst %.3f valu %.3f lds %.3f sca %.3f" % tuple(
    c.get(k, 0) / wc for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA")))
PY
