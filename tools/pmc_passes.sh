#!/bin/bash
# Collects PMC counters for the E-step kernels in separate rocprofv3 passes (kernel-trace/stats are NOT combined with --pmc).
# usage: tools/pmc_passes.sh <outdir> [bench args...]      (run on the GPU box from the repo root)
set -e
OUT=$1; shift
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-elbo-match --no-side-lines $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/p1 -- python3 $R/bench.py $ARGS > $R/$OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/$OUT/p2 -- python3 $R/bench.py $ARGS > $R/$OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/p3 -- python3 $R/bench.py $ARGS > $R/$OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/p4 -- python3 $R/bench.py $ARGS > $R/$OUT/p4.log 2>&1
find $R/$OUT -name "*.csv" | head
