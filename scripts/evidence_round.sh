#!/bin/bash
# Everything else that goes under profiles/ for a round, on the GPU box (after scripts/profile_round.sh):
#   bash scripts/evidence_round.sh r02_final
TAG=${1:-rXX}
OUT=gpurun_out
# (the round's bench line, ${TAG}_bench.json, is taken by scripts/profile_round.sh right after it has put the traffic record in place)
python3 scripts/bench_next_rows.py > $OUT/${TAG}_next_rows.txt 2>&1
for w in 1 2 4 8; do python3 scripts/rank_cost.py $w 2>&1 | grep -av amdgpu.ids | tail -14; done > $OUT/${TAG}_rank_cost.txt
python3 scripts/dist_overhead.py > $OUT/${TAG}_dist_overhead.txt 2>&1
python3 scripts/graph_overhead.py 2>&1 | grep -av amdgpu.ids > $OUT/${TAG}_graph_overhead.txt
bash scripts/train_e2e.sh B2 > $OUT/${TAG}_train_e2e.txt 2>&1
bash scripts/kstats_next_rows.sh ${TAG} > $OUT/${TAG}_next_rows_kernels.txt 2>&1
# (the timeline library is built BEFORE the run, where hipcc is known to work: bash scripts/build_abl.sh 9; conv_timeline.py
#  prints the source hash the library carries, so a stale one shows)
CMLPL_LIB=cmlpl_amd/libabl9.so python3 scripts/conv_timeline.py 2>&1 | grep -av amdgpu.ids > $OUT/${TAG}_timeline.txt
# round 5: a rank's shard of configs[2] (64 + 64 rows: one workgroup per CU) with the four-wave and the eight-wave kernels,
# and the eight-tile kernels of configs[4] (15 x 15 x 48) at its shard (8 + 64) and at 128 + 128
(for k in 0 1; do CMLPL_KS8=$k CMLPL_LIB=cmlpl_amd/libabl9.so python3 scripts/conv_timeline.py 103 9 64 64 2>&1 | grep -av "amdgpu.ids\|wgrad"; done
 CMLPL_TL_WIN=15 CMLPL_LIB=cmlpl_amd/libabl9.so python3 scripts/conv_timeline.py 48 20 8 64 2>&1 | grep -av "amdgpu.ids\|wgrad"
 CMLPL_TL_WIN=15 CMLPL_LIB=cmlpl_amd/libabl9.so python3 scripts/conv_timeline.py 48 20 128 128 2>&1 | grep -av "amdgpu.ids\|wgrad") > $OUT/${TAG}_timeline_shards.txt
# the general path's conv1 launches at the reference's own 20 x 20 x 60 windows (library built with CMLPL_STAMP_MIN_H=20)
[ -f cmlpl_amd/libabl9p.so ] && CMLPL_TL_WIN=20 CMLPL_LIB=cmlpl_amd/libabl9p.so python3 scripts/conv_timeline.py 60 16 128 128 2>&1 | grep -av amdgpu.ids > $OUT/${TAG}_timeline_p.txt
# whole-image inference from the cube (PaviaU-sized scene) + its kernel stats
(python3 scripts/bench_infer.py B2; python3 scripts/bench_infer.py B5) 2>&1 | grep -av amdgpu.ids > $OUT/${TAG}_infer.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf $OLDPWD/$OUT/${TAG}_infer_ks && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/${TAG}_infer_ks -o t -- python3 $OLDPWD/scripts/bench_infer.py B2 > /dev/null 2>&1)
cp $(find $OUT/${TAG}_infer_ks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_infer_kernel_stats.csv 2> /dev/null
# B5 per-rank costs with the eight-tile kernels and with the general kernels (CMLPL_FUSE_BIG=0), configs[2] shard with either width
(for k in 1 0; do echo "CMLPL_FUSE_BIG=$k"; CMLPL_FUSE_BIG=$k python3 scripts/rank_cost.py 8 B5 8 64 2>&1 | grep -av amdgpu.ids | tail -20; done
 for k in 0 1; do echo "CMLPL_KS8=$k"; CMLPL_KS8=$k python3 scripts/rank_cost.py 8 B3 64 64 2>&1 | grep -av amdgpu.ids | tail -12; done) > $OUT/${TAG}_rank_cost_variants.txt
for wl in P B4 B5; do python3 bench.py --workload $wl --steps 100 --no-cpu-baseline 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '%.4f ms/step' % d['ms_per_step'], '%.0f patches/s' % d['value'])"; done > $OUT/${TAG}_other_shapes.txt
python3 bench.py --workload B3 --gpus 1 --steps 100 --no-cpu-baseline 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B3 (512+512 on one GPU)', '%.4f ms/step' % d['ms_per_step'], '%.0f patches/s' % d['value'])" >> $OUT/${TAG}_other_shapes.txt
python3 bench.py --workload B5 --global-batch 64+512 --steps 100 --no-cpu-baseline 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B5 64+512 on one GPU', '%.4f ms/step' % d['ms_per_step'], '%.0f patches/s' % d['value'])" >> $OUT/${TAG}_other_shapes.txt
# strong-scaling rank costs (configs[2] over 1 / 2 / 4 / 8 ranks, configs[4] at 1 and 8), per-kernel times of a rank at W = 8
(for w in 1 2 4 8; do python3 scripts/rank_cost.py $w B3 $((512/w)) $((512/w)) 2>&1 | grep -av amdgpu.ids | tail -12; done
 python3 scripts/rank_cost.py 1 B5 64 512 2>&1 | grep -av amdgpu.ids | tail -20
 python3 scripts/rank_cost.py 8 B5 8 64 2>&1 | grep -av amdgpu.ids | tail -20) > $OUT/${TAG}_rank_cost_strong.txt
(echo "B2 128+128 rows per rank, W = 8"; bash scripts/kstats_rank.sh ${TAG}_w8 8 B2 128 128 pair_exp loss_ gemm
 echo "configs[2] 64+64 rows per rank, W = 8"; bash scripts/kstats_rank.sh ${TAG}_b3w8 8 B3 64 64 pair_exp loss_ gemm) > $OUT/${TAG}_rank_kernels.txt 2>&1
# machine floors the analysis leans on: launch-to-launch time of empty / small kernels, cold HBM read rate of the bank pattern
mkdir -p scripts/micro/bin
for m in launch_floor strided_read chunk_loop; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/$m.hip -o scripts/micro/bin/$m > /dev/null 2>&1 && timeout -k 5 120 scripts/micro/bin/$m
done > $OUT/${TAG}_machine_floors.txt 2>&1
bash scripts/pmc_instmix.sh ${TAG} > /dev/null 2>&1   # -> gpurun_out/TAG_instmix.txt
