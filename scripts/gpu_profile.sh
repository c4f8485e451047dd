#!/bin/bash
# rocprofv3 evidence for profiles/: (1) kernel trace + stats of the default benchmark, (2) PMC passes FETCH_SIZE and
# WRITE_SIZE (separate runs) for the HBM-side traffic per launch.  Summaries land in gpurun_out/ (copy into profiles/).
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02}
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_rd /tmp/prof_wr
CMD="python $R/bench.py --steps 2 --warmup 1 --precision fast --no-train --no-cpu-baseline --no-clock-power --no-live-traffic --no-videos30"   # (--precision fast: what auto picks on these weights, without the calibration's extra launches in the trace)
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o bench -- $CMD > $R/gpurun_out/${TAG}_kt_bench.json 2> $R/gpurun_out/${TAG}_kt.err
db=$(find /tmp/prof_kt -name "*.db" | head -1)
# idle gaps of the device inside the two timed steps (one patch-embedding launch per step since round 4: occurrences 1 .. 2 = one timed step)
python $R/scripts/rocpd_summary.py $db --gaps --gaps-window patch_embed 1 2 > $R/gpurun_out/${TAG}_kernel_trace_table.md 2>> $R/gpurun_out/${TAG}_kt.err
CMD1="python $R/bench.py --steps 1 --warmup 0 --precision fast --no-train --no-cpu-baseline --no-clock-power --no-live-traffic --no-videos30"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_rd -o rd -- $CMD1 > /dev/null 2> $R/gpurun_out/${TAG}_rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_wr -o wr -- $CMD1 > /dev/null 2> $R/gpurun_out/${TAG}_wr.err
rd=$(find /tmp/prof_rd -name "*.db" | head -1); wr=$(find /tmp/prof_wr -name "*.db" | head -1)
python $R/scripts/pmc_traffic.py $rd $wr > $R/gpurun_out/${TAG}_pmc_traffic.json 2>> $R/gpurun_out/${TAG}_rd.err
head -30 $R/gpurun_out/${TAG}_kernel_trace_table.md; cat $R/gpurun_out/${TAG}_pmc_traffic.json | head -40
# (3) one SQ-counter pass: where the waves of each kernel spend their cycles (scripts/pmc_sq.py)
rm -rf /tmp/prof_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES \
    -d /tmp/prof_sq -o sq -- $CMD1 > /dev/null 2> $R/gpurun_out/${TAG}_sq.err
sq=$(find /tmp/prof_sq -name "*.db" | head -1)
python $R/scripts/pmc_sq.py $sq > $R/gpurun_out/${TAG}_pmc_sq.md 2>> $R/gpurun_out/${TAG}_sq.err
head -24 $R/gpurun_out/${TAG}_pmc_sq.md
