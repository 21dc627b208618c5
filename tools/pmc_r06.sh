#!/bin/bash
# Round-6 counter passes of the launches that dominate the benchmarked graphs (batch 1 and 8), the
# self-attention launches included: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM
# section: FETCH_SIZE x2 on gfx950), one SQ pass (MFMA-busy, CU-busy, wait / issue-stall split) and a second
# SQ pass for the vector-ALU share.  Run on the GPU box from the repo root:
#   bash tools/pmc_r06.sh        -> gpurun_out/r06_pmc/{summary.json, *_counter_collection.csv}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_pmc
rm -rf $out; mkdir -p $out
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
SQ2="SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
while read SPEC; do
  [ -z "$SPEC" ] && continue
  i=$((i+1))
  for pass in FETCH_SIZE WRITE_SIZE SQ SQ2; do
    ctr=$pass; [ $pass = SQ ] && ctr="$SQ"; [ $pass = SQ2 ] && ctr="$SQ2"
    timeout 240 rocprofv3 --pmc $ctr --output-format csv -d $out/raw/${pass}_$i -o r -- python3 tools/pmc_gemm_probe.py $SPEC > $out/raw_${pass}_$i.log 2>&1
  done
  echo "$i $SPEC" >> $out/shapes.txt
done <<'SHAPES'
geglu 1024 10240 1280
lin 1024 1280 1280
lin 1024 1280 5120
lin 1024 3840 1280
conv 1 64 640 640
conv 1 32 1280 1280
attn 1 4096 640
attn 1 1024 1280
linattn 1 1024 1280
ln 1024 1280 1280
ln 4096 640 640
f16in 1024 1280 640
geglu 8192 10240 1280
lin 8192 1280 5120
lin 8192 3840 1280
conv 8 64 640 640
attn 8 4096 640
attn 8 1024 1280
SHAPES
python3 tools/pmc_r03_summary.py $out
# the sources the LOADED library was built from (its embedded hash): the provenance of these counters
python3 -c "import ctypes, mixdq_amd._C as C; f = C._lib.mixdq_build_csrc_sha16; f.restype = ctypes.c_char_p; print(f().decode())" > $out/csrc_sha16.txt 2>/dev/null
grep -l "rror" $out/raw_*.log 2>/dev/null | head -5 | while read f; do echo "== $f"; tail -3 $f; done
find $out/raw -name "*counter_collection.csv" | while read f; do cp $f $out/$(echo $f | sed 's|.*/raw/||; s|/.*||')_counter_collection.csv; done
rm -rf $out/raw
