# Re-measures only the HBM-side traffic of the body kernel (FETCH_SIZE / WRITE_SIZE passes, each in its own run) and
# rewrites gpurun_out/prof_<tag>/traffic.json with the digest of the kernel sources in the tree: for source changes
# that do not touch the kernel's memory behaviour.  usage (through gpurun): bash tools/collect_traffic.sh r02
set -x
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$set -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-teacher --no-create-data --no-trained-like > $O/pmc_$set.log 2>&1 || exit 1
  python $R/tools/pmc_summary.py $O/pmc_$set/*/*counter_collection.csv r2l_body > $O/pmc_$set.txt 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$set/*/*counter_collection.csv 'r2l_head' >> $O/pmc_$set.txt 2>&1
done
python $R/tools/traffic_json.py $O/pmc_FETCH_SIZE/*/*counter_collection.csv $O/pmc_WRITE_SIZE/*/*counter_collection.csv r2l_body fp16_fp8 $O/traffic.json
python $R/bench.py > $O/bench_n1_fp16_fp8.json 2> $O/bench.err
cat $O/traffic.json
