set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/p4b
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/p4b/stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 128 --k-eig 32 > gpurun_out/p4b/bench.json 2> gpurun_out/p4b/bench.err
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 128 --k-eig 32 > gpurun_out/p4b/b100.json 2> gpurun_out/p4b/b100.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 > gpurun_out/p4b/h20.json 2> gpurun_out/p4b/h20.err
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 > gpurun_out/p4b/h200.json 2> gpurun_out/p4b/h200.err
echo done
