#!/bin/bash
# PMC passes over the fp32 variant of the flag path next to the fp64 one (profiles/f32_time.py <workload> runs both):
# VALU instructions per wave of screen32_kernel against screen_kernel -- the instruction-count side of "why unpacked
# fp32 does not win" (DESIGN.md section 5).  usage: bash profiles/pmc_f32.sh <tag> <workload>
TAG=$1; W=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcf32_${TAG}_$W
mkdir -p $OUT
python3 $ROOT/profiles/fingerprint.py $ROOT > $OUT/kernel_fingerprint.txt
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o p -- python3 $ROOT/profiles/f32_time.py $W > $OUT/pmc_$name.log 2>&1 || { echo "pass $pass failed"; tail -3 $OUT/pmc_$name.log; exit 1; }
done
python3 $ROOT/profiles/summarize.py $OUT
