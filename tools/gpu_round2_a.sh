set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 build/valu_issue_microbench > gpurun_out/valu_issue.json 2> gpurun_out/valu_issue.err
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/calib_fetch -- $ROOT/build/fetch_calibration > $ROOT/gpurun_out/calib_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/calib_write -- $ROOT/build/fetch_calibration > $ROOT/gpurun_out/calib_write.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/calib_trace -- $ROOT/build/fetch_calibration > $ROOT/gpurun_out/calib_trace.log 2>&1)
find gpurun_out -name '*.csv' -size +8M -delete
timeout 2400 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
tail -30 gpurun_out/pytest_gpu.log
