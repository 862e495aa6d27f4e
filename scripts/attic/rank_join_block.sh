# the rank-of-8 replay at 1 M rows: the masked join's work units at their automatic width and at 64 ("join_masked_block")
mkdir -p gpurun_out/r06k
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 scripts/rank_of_8.py record --rows 1000000 --world 8 --rank 3 --dir /dev/shm/r8 > gpurun_out/r06k/record.txt 2>&1
timeout 300 python scripts/rank_of_8.py replay --rows 1000000 --world 8 --rank 3 --dir /dev/shm/r8 --reps 7 --check --out gpurun_out/r06k/rank3_of_8_1M.json 2>&1 | grep "rank_ms\"\|knn_mfma_join_f16\"\|one_gpu\|equals"
timeout 300 python scripts/rank_of_8.py replay --rows 1000000 --world 8 --rank 3 --dir /dev/shm/r8 --reps 3 --rank-only --opt join_masked_block=64 2>&1 | grep "rank_ms\"\|knn_mfma_join_f16\""
