# end-to-end eval rollouts through imitate_episodes.py (SyntheticEnv, 3 cams, temporal ensembling, incl. H2D / D2H) + the eval-shard bench mode
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rollout_r03
mkdir -p $O
cd $R/act-plus-plus_amd
COMMON="--eval --task_name sim_transfer_cube_scripted --ckpt_dir $O/ck --policy_class ACT --kl_weight 10 --chunk_size 100 --hidden_dim 512 --batch_size 8 --dim_feedforward 3200 --num_steps 1 --lr 1e-5 --seed 0 --temporal_agg --synthetic_env"
python3 imitate_episodes.py $COMMON --num_rollouts 50 --max_batch 50 > $O/eval_rollout_50ep.log 2>&1
python3 imitate_episodes.py $COMMON --num_rollouts 2 --max_batch 1 > $O/eval_rollout_b1.log 2>&1
cd $R
python3 bench.py --mode eval-shard --episode-len 60 > $O/bench_evalshard.json 2> $O/bench_evalshard.err
tail -3 $O/eval_rollout_50ep.log; tail -3 $O/eval_rollout_b1.log; tail -c 600 $O/bench_evalshard.json
