#!/bin/bash
# Collects the per-round evidence on the GPU box in one call: bench lines of every workload, the rocprofv3 kernel trace
# and the PMC passes of the headline workload, under gpurun_out/v7r/ (or $EVIDENCE_DIR) (copied into profiles/ by hand afterwards).
set -e
R=$PWD
O=gpurun_out/${EVIDENCE_DIR:-v7r}
mkdir -p $O
python bench.py > $O/ns_bench.json 2> $O/ns_bench.err
echo ns done
python bench.py --workload c2 > $O/c2_bench.json 2> $O/c2.err
python bench.py --workload c3 > $O/c3_bench.json 2> $O/c3.err
echo c3 done
python bench.py --workload c1 > $O/c1_bench.json 2> $O/c1.err
python bench.py --workload c5 --steps 5 --warmup 2 > $O/c5_bench.json 2> $O/c5.err
python bench.py --workload c5s --steps 5 --warmup 2 > $O/c5s_bench.json 2> $O/c5s.err
echo c5 done
python bench.py --model white --steps 10 --warmup 3 --no-cpu-baseline > $O/ns_white_bench.json 2> $O/white.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-elbo-match --no-side-lines > $R/$O/kt.log 2>&1
cd $R
echo kt done
bash tools/pmc_passes.sh $O/pmc > $O/pmc.log 2>&1
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1 || true
echo all done
