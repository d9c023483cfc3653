set -o pipefail
mkdir -p gpurun_out/c5
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --share-gpu --no-class-surface > gpurun_out/c5/r06_bench_survey3_65536_4ranks_share_gpu.json 2> gpurun_out/c5/share4.err; echo "share4 rc=$?"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --share-gpu --no-class-surface > gpurun_out/c5/r06_bench_survey3_65536_2ranks_share_gpu.json 2> gpurun_out/c5/share2.err; echo "share2 rc=$?"
timeout -k 10 200 python bench.py --force-dist --no-class-surface > gpurun_out/c5/r06_bench_survey3_65536_force_dist.json 2> gpurun_out/c5/force.err; echo "force rc=$?"
timeout -k 10 200 python bench.py --node 2 --share-gpu --no-class-surface > gpurun_out/c5/r06_bench_survey3_65536_node2_share_gpu.json 2> gpurun_out/c5/node2.err; echo "node2 rc=$?"
timeout -k 10 300 python bench.py --workload corpus --corpus-streams 600 --no-class-surface > gpurun_out/c5/r06_bench_corpus_600.json 2> gpurun_out/c5/corpus600.err; echo "corpus600 rc=$?"
