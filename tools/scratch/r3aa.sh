set -e
OUT=gpurun_out/r3aa
mkdir -p $OUT
rm -f $OUT/gpu_measured_errors.json
FETA_RECORD_ERRORS=$OUT/gpu_measured_errors.json python -m pytest tests -m gpu -x -q > $OUT/tests_record.log 2>&1 || { tail -30 $OUT/tests_record.log; exit 1; }
tail -1 $OUT/tests_record.log
cp $OUT/gpu_measured_errors.json tests/golden/gpu_measured_errors.json
python -m pytest tests -m gpu -x -q > $OUT/tests_guard.log 2>&1 || { tail -30 $OUT/tests_guard.log; exit 1; }
tail -1 $OUT/tests_guard.log
# second recording in another process order: are the errors reproducible?
FETA_RECORD_ERRORS=$OUT/gpu_measured_errors_2.json python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > $OUT/tests_record2.log 2>&1
python - <<P
import json
a=json.load(open('$OUT/gpu_measured_errors.json'))['errors']; b=json.load(open('$OUT/gpu_measured_errors_2.json'))['errors']
worst=max((b[k]/max(a[k],1e-12) for k in b if k in a), default=0)
print(len(a), 'recorded;', len(b), 're-recorded; worst ratio', worst)
P
