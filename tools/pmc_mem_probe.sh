#!/bin/bash
# usage: tools/pmc_mem_probe.sh <kernel-substring> <probe.py> -- memory-path counters of one kernel (TA / TCP / TCC; separate
# passes, --kernel-trace only): where a gather-bound kernel waits
cd /root/repo
export TMPDIR=/tmp
K=$1; P=$2
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_BUSY_avr TCC_TAG_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc/$tag -- python /root/repo/$P > /dev/null 2>/root/repo/gpurun_out/pmc/$tag.err)
  f=$(find gpurun_out/pmc/$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$K" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
acc = collections.defaultdict(list)
for r in rows:
    acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    v = v[1:] if len(v) > 1 else v
    print("%-62s %-40s n=%d mean=%.4g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
done
rm -rf gpurun_out/pmc
