#!/bin/bash
# Round 6, visit T: what the chained iSTFT and the forward STFT wait for -- memory-pipeline counters of the --steps 20 launch
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/r06_t_avail.txt 2>&1 || rocprofv3 -L > $OUT/r06_t_avail.txt 2>&1
grep -o "\b\(TA\|TCP\|TCC\|TD\|SQ\|SQC\|GRBM\)_[A-Za-z0-9_]*" $OUT/r06_t_avail.txt | sort -u > $OUT/r06_t_names.txt; wc -l $OUT/r06_t_names.txt
CMD="python bench.py --steps 20 --warmup 5 --legs= --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 --no-parity-check --max-rounds 60"
n=0
while read -r line; do
  n=$((n+1))
  rm -rf $OUT/pmc_t$n
  timeout 300 rocprofv3 --kernel-trace --pmc $line --output-format csv -d $OUT/pmc_t$n -o p -- $CMD > /dev/null 2> $OUT/pmc_t$n.err; echo "pass $n ($line) exit $?"
done <<'LIST'
TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum
TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES
SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA
LIST
python - <<'PY' > $OUT/r06_t_counters.txt
import csv, glob, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_t*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = None
        for k in ("istft_chain_kernel", "stft_forward_wave_kernel", "final_bf16x3_kernel", "deconv2_stream_bf16_kernel"):
            if k in name: key = k
        if key is None: continue
        agg[(key, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-44s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUT/r06_t_counters.txt | head -150
find $OUT -path "*pmc_t*" -name "*.csv" -delete; find $OUT -name "*.db" -delete; rm -rf $OUT/pmc_t*
