# Copies the summaries a `gpurun -- bash tools/refresh_profiles.sh` call left under gpurun_out/refresh/ into profiles/
# with the round's prefix: bash tools/collect_profiles.sh r03
P=${1:?round prefix, e.g. r03}
O=gpurun_out/refresh
cp $O/bench.json profiles/${P}_bench.json
cp $O/bench_driver_cmd.json profiles/${P}_bench_driver_cmd.json
for prec in bf16 fp32; do
  cp $O/ks_$prec/ks_kernel_stats.csv profiles/${P}_kernel_stats_$prec.csv
  cp $O/pmc_$prec.txt profiles/${P}_pmc_$prec.txt
done
cp $O/ks_train/ks_kernel_stats.csv profiles/${P}_kernel_stats_train.csv
cp $O/attn_traffic.json profiles/attn_traffic.json
cp $O/stage_ms.txt profiles/${P}_stage_ms.txt
cp $O/config_sweep.txt profiles/${P}_config_sweep.txt
cp $O/block_size_sweep.txt profiles/${P}_block_size_sweep.txt
cp $O/shard_overhead.txt profiles/${P}_shard_overhead.txt
cp $O/sort_and_combine_micro.txt profiles/${P}_sort_and_combine_micro.txt
grep -v amdgpu.ids $O/attn_block.txt > profiles/${P}_attn_block_and_train.txt
grep -v amdgpu.ids $O/model_and_prepare.txt > profiles/${P}_model_and_prepare.txt
grep -v amdgpu.ids $O/host_overhead.txt > profiles/${P}_host_overhead.txt
{
  for how in plain p2p rccl; do echo "== 3 tables, $how"; grep -v amdgpu.ids $O/timeline_$how.txt; done
  for how in plain p2p p2pview rccl; do echo "== 1 table (BASELINE config 4 on one rank), $how"; grep -v amdgpu.ids $O/timeline_T1_$how.txt; done
} > profiles/${P}_timelines.txt
