#!/bin/bash
# usage: tools/pmc_probe.sh <kernel-substring> <probe.py> -- SQ counters of one kernel (separate passes, --kernel-trace only)
cd /root/repo
export TMPDIR=/tmp
K=$1; P=$2
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_INST_CYCLES_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc/$tag -- python /root/repo/$P > /dev/null 2>/root/repo/gpurun_out/pmc/$tag.err)
  f=$(find gpurun_out/pmc/$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$K" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
acc = collections.defaultdict(list)
for r in rows:
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    v = v[1:] if len(v) > 1 else v
    print("%-28s n=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
  tail -2 gpurun_out/pmc/$tag.err | cut -c1-200
done
rm -rf gpurun_out/pmc
