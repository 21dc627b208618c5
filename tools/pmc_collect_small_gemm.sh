#!/bin/bash
# SQ / TA / TCP counter passes for the small-tile igemm in its two regimes (one workgroup per CU,
# 2.5 per CU).  Run on the GPU box from the repo root; writes gpurun_out/pmc_small/<pass>_<shape>/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
C="TA_BUSY TA_TOTAL_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCR_TCP_STALL_CYCLES"
for pass in A B; do   # pass C (TA / TCP) aborts inside rocprofv3 on this image: left out
  for shape in "1024 1024 5120" "2048 1280 5120"; do
    tag=$(echo $shape | tr ' ' x)
    timeout 120 rocprofv3 --pmc ${!pass} --output-format csv -d gpurun_out/pmc_small/${pass}_${tag} -o r -- python3 tools/pmc_gemm_probe.py $shape 4 > gpurun_out/pmc_small_${pass}_${tag}.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_small/*")):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f:
        print(d, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "igemm_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[-1], {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
