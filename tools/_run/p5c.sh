set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/p5c
F4="--no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 128 --k-eig 32"
F5="--no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape molhiv --batch 1024 --n-pad 64 --dtype bf16"
python3 bench.py --steps 200 --warmup 20 $F4 > gpurun_out/p5c/c4.json 2> gpurun_out/p5c/c4.err
python3 bench.py --steps 100 --warmup 20 $F5 > gpurun_out/p5c/c5.json 2> gpurun_out/p5c/c5.err
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/p5c/s4 -- python3 bench.py --steps 30 --warmup 5 $F4 > gpurun_out/p5c/b4.json 2> gpurun_out/p5c/b4.err
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/p5c/s5 -- python3 bench.py --steps 30 --warmup 5 $F5 > gpurun_out/p5c/b5.json 2> gpurun_out/p5c/b5.err
echo done
