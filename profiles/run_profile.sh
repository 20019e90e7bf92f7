#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py's default workload.
# usage: bash profiles/run_profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 $ROOT/profiles/fingerprint.py $ROOT > $OUT/kernel_fingerprint.txt
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-planner --no-edges --no-legs --no-distance --no-fp32 $@"
# trace pass 1, --one-ctx: the launches back to back on one stream -- the launch duration roofline.achieved is priced on
# (bench.py measures the same live, roofline.kernel_ms).  Trace pass 2, the default command: its timed steps also go
# through the pipe, where two contexts' kernels share the chip and a kernel's duration includes that shared time.
rm -rf $OUT/trace $OUT/trace_pipe
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py $ARGS --one-ctx > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipe -o t -- python3 $ROOT/bench.py $ARGS > $OUT/trace_pipe.log 2>&1 || { tail -5 $OUT/trace_pipe.log; exit 1; }
if [ -n "$MVK_PROFILE_TRACE_ONLY" ]; then exit 0; fi   # (the PMC passes of this kernel fingerprint are already there)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  # (counters do not depend on the clocks: no pre-warm steps in the PMC passes, which serialise every dispatch)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o p -- python3 $ROOT/bench.py $ARGS --prewarm 0 > $OUT/pmc_$name.log 2>&1 || { echo "pass $pass failed"; tail -3 $OUT/pmc_$name.log; }
done
python3 $ROOT/profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# afterwards, here: python3 profiles/summarize.py gpurun_out/prof_<tag> --merge profiles/latest_pmc.json --workload <wl> --source profiles/<round>/<wl>_summary.txt
