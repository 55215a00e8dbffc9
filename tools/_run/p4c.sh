set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/p4c
F="--no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 128 --k-eig 32"
python3 bench.py --steps 200 --warmup 20 $F > gpurun_out/p4c/a.json 2> gpurun_out/p4c/a.err
python3 bench.py --steps 200 --warmup 20 $F --layer-norm --no-pe > gpurun_out/p4c/ln.json 2> gpurun_out/p4c/ln.err
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/p4c/stats -- python3 bench.py --steps 30 --warmup 5 $F > gpurun_out/p4c/bench.json 2> gpurun_out/p4c/bench.err
python3 bench.py --steps 200 --warmup 20 $F > gpurun_out/p4c/b.json 2> gpurun_out/p4c/b.err
echo done
