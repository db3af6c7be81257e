#!/bin/bash
# copy the judged summaries of a validation run into profiles/ (run here, not on the GPU box):
#   bash tools/collect_profiles.sh r02v3 [r02pmc] [timeline tag] [sq tag]
V=gpurun_out/${1:?validation tag}; P=gpurun_out/${2:-r02pmc}; T=gpurun_out/${3:-none}; S=gpurun_out/${4:-none}
for m in infer infer_2rank_gloo infer_bf16 train_bf16 train_bf16_strong train_f32 gan_bf16 gan_f32 centroids weightmap weightmap2 frontend; do
  [ -s $V/bench_$m.json ] && cp $V/bench_$m.json profiles/r02_bench_$m.json
done
for m in infer train gan; do
  f=$(find $V/prof_$m -name "*kernel_stats.csv" | head -1)
  case $m in infer) o=r02_infer_kernel_stats.csv;; train) o=r02_train_bf16_kernel_stats.csv;; gan) o=r02_gan_bf16_kernel_stats.csv;; esac
  [ -n "$f" ] && cp $f profiles/$o
done
for f in r02_pmc_train_bf16_traffic r02_pmc_gan_bf16_traffic r02_pmc_hbm_traffic; do [ -s $P/$f.json ] && cp $P/$f.json profiles/$f.json; done
[ -s $T/step_timeline.txt ] && cp $T/step_timeline.txt profiles/r02_train_bf16_step_timeline.txt
[ -d $S ] && python tools/pmc_sq_summary.py $S > profiles/r02_pmc_sq_summary.json
ls -la profiles | grep r02_ | awk '{print $5, $9}'
