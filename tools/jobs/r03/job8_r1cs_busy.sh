#!/bin/bash
# tools/r03/job8_r1cs_busy.sh — run ON THE GPU BOX: how busy the stream kernel keeps the CU's units (derived counters, one pass each)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_job8
mkdir -p $out
for c in compression nova_vesta; do
  for pmc in "VALUBusy SALUBusy" "MemUnitBusy LDSBankConflict" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    rocprofv3 --pmc $pmc --output-format csv -d $out/${c}_$tag -- python3 tools/ubench/r1cs_profile_target.py $c > $out/${c}_$tag.log 2>&1 || { echo "$c $tag failed"; tail -3 $out/${c}_$tag.log; continue; }
    f=$(find $out/${c}_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$c" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "stream" in r["Kernel_Name"] or "deferred" in r["Kernel_Name"]:
        k = "stream" if "stream" in r["Kernel_Name"] else "deferred"
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, cn), v in sorted(acc.items()):
    print(f"{sys.argv[2]:12s} {k:9s} {cn:24s} mean {sum(v)/len(v):.6g}  (n={len(v)})")
PY
    rm -rf $out/${c}_$tag
  done
done
