set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/p4d
python3 -m pytest tests/test_kernels_gpu.py -x -q -k coeff > gpurun_out/p4d/test.txt 2>&1
F4="--no-cpu-baseline --no-literal --stream-batch 0 --kernel-iters 1 --shape pattern --batch 64 --n-pad 128 --k-eig 32"
python3 bench.py --steps 200 --warmup 20 $F4 > gpurun_out/p4d/c4.json 2> gpurun_out/p4d/c4.err
FETA_COEFF_WIDE=0 python3 bench.py --steps 200 --warmup 20 $F4 > gpurun_out/p4d/c4old.json 2> gpurun_out/p4d/c4old.err
python3 bench.py --steps 200 --warmup 20 $F4 --layer-norm --no-pe > gpurun_out/p4d/c4ln.json 2> gpurun_out/p4d/c4ln.err
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/p4d/s4 -- python3 bench.py --steps 30 --warmup 5 $F4 > gpurun_out/p4d/b4.json 2> gpurun_out/p4d/b4.err
echo done
