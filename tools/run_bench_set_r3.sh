set -e
O=gpurun_out/r3n; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/ns_bench.json 2> $O/ns_bench.err
echo "ns done"; tail -c 600 $O/ns_bench.json | head -c 300; echo
python bench.py --workload c2 --steps 20 --warmup 5 > $O/c2_bench.json 2> $O/c2_bench.err; echo "c2 done"
python bench.py --workload c3 --steps 20 --warmup 5 > $O/c3_bench.json 2> $O/c3_bench.err; echo "c3 done"
python bench.py --rows 125000 --steps 20 --warmup 5 --no-side-lines > $O/ns125k_bench.json 2> $O/ns125k_bench.err; echo "125k done"
python bench.py --workload c1 --steps 20 --warmup 5 > $O/c1_bench.json 2> $O/c1_bench.err; echo "c1 done"
