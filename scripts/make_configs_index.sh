#!/bin/bash
# scripts/make_configs_index.sh -- profiles/r02_all_configs.md from the logs under profiles/r02_logs/
L=profiles/r02_logs
{
echo "# Round 2 -- every configuration quoted in DESIGN.md, one bench.py line each (raw logs: profiles/r02_logs/; made by scripts/make_configs_index.sh)"
echo; echo "## final code of the round"; echo
python scripts/configs_table.py \
  "default: configs[2] per-GPU shard (256 games x 800 sims, 10x128 fp32), 20 timed plies=$L/r04a_bench_default.log" \
  "same, 200 timed plies=$L/r04a_bench_200a.log" "same, 200 timed plies (2nd run)=$L/r04a_bench_200b.log" \
  "same, 600 timed plies=$L/r04a_bench_steady600.log" "same, 5000 timed plies (3847 games finished, mean 331 plies)=$L/r05c_bench_5000.log" "same, 60 timed plies=$L/r04j_steps60.log" \
  "configs[1] (256 games x 400 sims)=$L/r04a_bench_cfg1.log" "configs[4] per-GPU shard (512 games, 20x256 fp16)=$L/r04i_bench_cfg4.log" "configs[4] shard before the head kernels took fp16 planes=$L/r04a_bench_cfg4.log" \
  "2048 games per GPU (10x128 fp32)=$L/r04a_bench_g2048.log" \
  "N=1 through RCCL (--force-dist, exchange every 4 plies, 60 timed plies)=$L/r04j_bench_nccl1.log" "N=1 through RCCL with 4 hardware queues (ROCm default)=$L/r04h_bench_nccl1.log" \
  "2 ranks sharing the GPU, gloo, 128 games each, exchange every 4 plies=$L/r04j_bench_dist2_gloo.log" \
  "fast mode, 256 games x 16 leaves, fp32 net=$L/r04b_bench_fast.log" "fast mode, 256 games x 16 leaves, fp16 net=$L/r04b_bench_fast_f16.log" \
  "fast mode, 4096 games x 4 leaves, fp16 net=$L/r04b_bench_fast_4096.log"
echo; echo "UCI latency path (configs[3], \`r04c_uci_latency.log\`): $(tail -1 $L/r04c_uci_latency.log)"
echo; echo "GPU tests of the final code: \`r04a_gpu_tests.log\` (90 passed, 2 skipped)."
echo; echo "## earlier in the round (kept for the history in DESIGN.md section 5)"; echo
python scripts/configs_table.py \
  "mid-round (before the head kernels, the tower's triple buffering and the lazy begin): default=$L/bench_default.log" \
  "mid-round: 200 timed plies=$L/bench_200_steps_a.log" "mid-round: configs[1]=$L/bench_cfg1.log" "mid-round: configs[4] shard=$L/bench_cfg4.log" \
  "mid-round: 2048 games=$L/bench_g2048.log" "mid-round: N=1 through RCCL=$L/bench_nccl1.log" "mid-round: fast, fp32 net=$L/bench_fast.log" "mid-round: fast, fp16 net=$L/bench_fast_f16.log" \
  "mid-round: fast 32768 games x 1 leaf x 64 sims, 4x64 fp16 net (many-trees roofline point)=$L/bench_fast_32768_games.log" \
  "start of the round's step-kernel / host-path / tower work: 600 timed plies=$L/bench_600_steps_before_step_kernel_work.log"
} > profiles/r02_all_configs.md
