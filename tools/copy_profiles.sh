#!/bin/bash
# Copies what is to be judged from a tools/prof_round.sh result (gpurun_out/<tag>/, scratch) into profiles/ (tracked).
#   tools/copy_profiles.sh r4a round4_c
T=$1; P=profiles/$2
O=gpurun_out/$T
cp $O/bench.json ${P}_bench.json
cp $O/layer_table.txt ${P}_layer_table.txt
cp $O/rp/r1_kernel_stats.csv ${P}_kernel_stats.csv
cp $O/rp_lanes/r1_kernel_stats.csv ${P}_kernel_stats_lanes.csv
cp $O/hbm_traffic.json ${P}_hbm_traffic.json
cp $O/bench_force_dist.json ${P}_bench_force_dist_one_rank_rccl.json
cat $O/reward_latency.txt $O/reward.txt > ${P}_reward_calls.txt
cp $O/real.txt ${P}_context_ae_real.txt
cp $O/real_layers.txt ${P}_context_ae_real_layers.txt
cp $O/config4.txt ${P}_config4_inception_end_to_end.txt
cp $O/frontend_layers.txt ${P}_inception_frontend_layers.txt
grep -E "passed|failed" $O/pytest_gpu.txt > ${P}_pytest_gpu.txt
ls $O/pmc_wconvt_kernel/*.txt 2>/dev/null | head -1 | xargs -I{} cp {} ${P}_pmc_wconvt.txt
ls -la ${P}_* | awk '{print $5, $9}'
