#!/bin/bash
# copy the judged summaries of a validation run into profiles/ (run here, not on the GPU box):
#   RND=r03 bash tools/collect_profiles.sh r03v1 [r03pmc] [timeline tag] [sq tag]
RND=${RND:-r02}
V=gpurun_out/${1:?validation tag}; P=gpurun_out/${2:-${RND}pmc}; T=gpurun_out/${3:-none}; S=gpurun_out/${4:-none}
for m in infer infer_2rank_gloo infer_bf16 train_bf16 train_bf16_strong train_f32 gan_bf16 gan_f32 centroids weightmap weightmap2 frontend; do
  [ -s $V/bench_$m.json ] && cp $V/bench_$m.json profiles/${RND}_bench_$m.json
done
for m in infer train gan; do
  f=$(find $V/prof_$m -name "*kernel_stats.csv" | head -1)
  case $m in infer) o=${RND}_infer_kernel_stats.csv;; train) o=${RND}_train_bf16_kernel_stats.csv;; gan) o=${RND}_gan_bf16_kernel_stats.csv;; esac
  [ -n "$f" ] && cp $f profiles/$o
done
for f in pmc_train_bf16_traffic pmc_gan_bf16_traffic pmc_hbm_traffic; do [ -s $P/${RND}_$f.json ] && cp $P/${RND}_$f.json profiles/${RND}_$f.json; done
[ -s $T/step_timeline.txt ] && cp $T/step_timeline.txt profiles/${RND}_train_bf16_step_timeline.txt
[ -d $S ] && python tools/pmc_sq_summary.py $S > profiles/${RND}_pmc_sq_summary.json
ls -la profiles | grep ${RND}_ | awk '{print $5, $9}'
