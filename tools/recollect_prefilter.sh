set -u
tag=${1:-r06}; out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
rm -rf $out/pmc_fetch_pf $out/pmc_write_pf $out/pmc_mfma_pf
P="python3 tools/prefilter_bench.py --queries 1,64 --steps 3 --both"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_pf -o f -- $P > $out/pmc_fetch_pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_pf -o w -- $P > $out/pmc_write_pf.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $out/pmc_mfma_pf -o m -- $P > $out/pmc_mfma_pf.log 2>&1
python3 tools/pmc_summary.py $out $out/${tag}_pmc_prefilter _pf > $out/${tag}_pmc_prefilter.log 2>&1
tail -4 $out/${tag}_pmc_prefilter.log
rm -rf $out/lat
rocprofv3 --kernel-trace -d $out/lat -o t -- python3 tools/latency_bench.py 1x1 > $out/lat.log 2>&1
python3 tools/latency_trace.py $out/lat/t_results.db > $out/${tag}_latency_trace.txt 2>&1
cp $out/${tag}_pmc_prefilter.json $out/${tag}_pmc_prefilter_counters.csv profiles/
python3 bench.py > $out/${tag}_bench_line.json 2> $out/${tag}_bench.err
python3 -c "
import json; r=json.load(open('$out/${tag}_bench_line.json'))
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline'].get('traffic_over_algorithmic'), r['roofline']['prefilter']['frac'], r['roofline']['prefilter'].get('traffic_over_algorithmic'), r['roofline']['prefilter_q64']['frac'])
print(r['dropin_forward']['ms_per_batch'], r['dropin_forward']['executed_rate_vs_arena_path'], r['localizer']['ms_per_query'], r['latency_config1']['ms_per_query'], r['latency_config1']['hip_graph']['ms_per_query'], [r['shard_proxy_%d'%w]['projected_efficiency'] for w in (2,4,8)])
"
rm -rf $out/pmc_fetch_pf $out/pmc_write_pf $out/pmc_mfma_pf $out/lat
