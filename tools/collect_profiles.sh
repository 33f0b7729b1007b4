#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run from the repo root through gpurun):
#   tools/collect_profiles.sh r04
# writes gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards.  Counters are collected in their own
# passes (never together with API traces), each a separate run of the same command, the program directly after `--`.
set -u
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
rm -rf $out/prof_$tag $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/pmc_fetch_pf $out/pmc_write_pf $out/pmc_mfma_pf
# (1) per-kernel time of the whole default command (extras included: the pre-filter / config-5 / latency kernels)
rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o t -- python3 bench.py --steps 3 --warmup 1 --cpu_queries 0 \
    > $out/${tag}_prof_bench.json 2> $out/prof_$tag.log
python3 tools/rocpd_summary.py $out/prof_$tag/t_results.db > $out/${tag}_bench_kernel_stats_all.csv 2>> $out/prof_$tag.log
# (2) the timed region only: the last launches of the trace (3 steps), so that per-kernel averages are those of the step
rm -rf $out/prof_${tag}_step
rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_step -o t -- python3 bench.py --steps 3 --warmup 1 --cpu_queries 0 \
    --no_extras > $out/${tag}_prof_step.json 2>> $out/prof_$tag.log
win=$(python3 -c "import json,sys; print(3 * json.load(open('$out/${tag}_prof_step.json'))['ms_per_step'] + 0.5)")
python3 tools/rocpd_summary.py $out/prof_${tag}_step/t_results.db $win > $out/${tag}_bench_kernel_stats.csv 2>> $out/prof_$tag.log
python3 tools/rocpd_summary.py $out/prof_${tag}_step/t_results.db $win --gaps > $out/${tag}_step_gaps.txt 2>> $out/prof_$tag.log
# (3) the opt-in split-bf16 path, same window
rm -rf $out/prof_${tag}_split
rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_split -o t -- python3 bench.py --steps 3 --warmup 1 --cpu_queries 0 \
    --no_extras --set_option split_bf16=1 > $out/${tag}_prof_split.json 2>> $out/prof_$tag.log
win=$(python3 -c "import json,sys; print(3 * json.load(open('$out/${tag}_prof_split.json'))['ms_per_step'] + 0.5)")
python3 tools/rocpd_summary.py $out/prof_${tag}_split/t_results.db $win > $out/${tag}_bench_kernel_stats_split.csv 2>> $out/prof_$tag.log
# (4) PMC passes over the step
B="python3 bench.py --steps 2 --warmup 1 --cpu_queries 0 --no_extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- $B > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- $B > $out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv \
    -d $out/pmc_mfma -o m -- $B > $out/pmc_mfma.log 2>&1
python3 tools/pmc_summary.py $out $out/${tag}_pmc > $out/${tag}_pmc.log 2>&1
tail -3 $out/${tag}_pmc.log
# (5) PMC passes over the MAD-scale pre-filter (BASELINE configs[2]; 1 and 64 queries in one process; --split_bf16: the opt-in
#     three-piece bf16 kernel runs too, for 64 queries, in the same process)
P="python3 tools/prefilter_bench.py --queries 1,64 --steps 3 --both"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_pf -o f -- $P > $out/pmc_fetch_pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_pf -o w -- $P > $out/pmc_write_pf.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv \
    -d $out/pmc_mfma_pf -o m -- $P > $out/pmc_mfma_pf.log 2>&1
python3 tools/pmc_summary.py $out $out/${tag}_pmc_prefilter _pf > $out/${tag}_pmc_prefilter.log 2>&1
tail -8 $out/${tag}_pmc_prefilter.log
# (6) instruction counts by class + MFMA / vector co-execution over the step: the issue-time model (tools/pmc_issue.py)
rm -rf $out/pmc_inst $out/pmc_coexec
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM \
    --output-format csv -d $out/pmc_inst -o i -- $B > $out/pmc_inst.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d $out/pmc_coexec -o c -- $B > $out/pmc_coexec.log 2>&1
python3 tools/pmc_issue.py $out $out/${tag}_pmc_issue.csv > $out/${tag}_pmc_issue.log 2>&1
cat $out/${tag}_pmc_issue.log
# (7) the opt-in LDS-resident decoder cross-attention (dec_fold=4) next to the default two-read form: FETCH_SIZE per launch
rm -rf $out/pmc_fetch_dc
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_dc -o f -- python3 tools/dec_cross_bench.py 20000 3 \
    > $out/pmc_fetch_dc.log 2>&1
python3 tools/fetch_per_kernel.py $out/pmc_fetch_dc/f_counter_collection.csv dec_cross > $out/${tag}_dec_cross_fetch.txt 2>&1
cat $out/${tag}_dec_cross_fetch.txt
# (8) the ragged split (config2_ragged): the step's GPU-idle gaps -- no host sync inside a step
rm -rf $out/prof_${tag}_ragged
rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_ragged -o t -- python3 tools/ragged_step.py 4 > $out/${tag}_ragged_step.json 2>> $out/prof_$tag.log
win=$(python3 -c "import json; print(3 * json.load(open('$out/${tag}_ragged_step.json'))['ms_per_step'] + 0.5)")
python3 tools/rocpd_summary.py $out/prof_${tag}_ragged/t_results.db $win --gaps > $out/${tag}_ragged_step_gaps.txt 2>> $out/prof_$tag.log
# (9) the launch sequence of one single-query step (BASELINE configs[0])
rm -rf $out/lat
rocprofv3 --kernel-trace -d $out/lat -o t -- python3 tools/latency_bench.py 1x1 > $out/lat.log 2>&1
python3 tools/latency_trace.py $out/lat/t_results.db > $out/${tag}_latency_trace.txt 2>&1
# (10) the reference's own entry (INTEGRATION Option A): CONE.forward + forward_clip_matching on the 640-window padded batch
rm -rf $out/dropin
rocprofv3 --kernel-trace --stats -d $out/dropin -o t -- python3 tools/dropin_bench.py dropin 20 > $out/${tag}_dropin.json 2> $out/dropin.log
python3 tools/dropin_trace.py $out/dropin/t_results.db scan_lengths_kernel 6 > $out/${tag}_dropin_trace.txt 2>&1
# (11) rank 0's share of the 8-rank window-sharded split (no collective), one step in flight: the launch sequence of one step
rm -rf $out/proxy8
rocprofv3 --kernel-trace -d $out/proxy8 -o t -- python3 tools/proxy_bench.py 8 0 10 > $out/${tag}_proxy8.json 2> $out/proxy8.log
python3 tools/step_trace.py $out/proxy8/t_results.db > $out/${tag}_proxy8_trace.txt 2>&1
python3 tools/rocpd_summary.py $out/proxy8/t_results.db 60 > $out/${tag}_proxy8_kernel_stats.csv 2>> $out/proxy8.log
# (12) other slot counts / pre-norm / --use_txt_pos: ms per step
python3 tools/slots_step.py 5 10 8 3 16 > $out/${tag}_slots_step.txt 2>&1
python3 tools/prenorm_step.py > $out/${tag}_prenorm_step.txt 2>&1
python3 tools/txtpos_step.py > $out/${tag}_txtpos_step.txt 2>&1
# the bench line last: roofline.traffic is read from profiles/<tag>_pmc_*.json of THIS collection
cp $out/${tag}_pmc_traffic.json $out/${tag}_pmc_counters.csv $out/${tag}_pmc_prefilter.json $out/${tag}_pmc_prefilter_counters.csv profiles/
python3 bench.py > $out/${tag}_bench_line.json 2> $out/${tag}_bench.err
cat $out/${tag}_bench_line.json
# gpurun copies back at most 64 MiB: the rocpd databases and raw counter tables stay on the box, their summaries travel
mkdir -p $out/${tag}_profiles
cp $out/${tag}_*.csv $out/${tag}_*.json $out/${tag}_*.txt $out/${tag}_profiles/ 2>/dev/null
rm -rf $out/prof_$tag $out/prof_${tag}_step $out/prof_${tag}_split $out/pmc_fetch $out/pmc_write $out/pmc_mfma \
       $out/pmc_fetch_pf $out/pmc_write_pf $out/pmc_mfma_pf $out/pmc_inst $out/pmc_coexec $out/pmc_fetch_dc \
       $out/prof_${tag}_ragged $out/lat $out/dropin $out/proxy8
