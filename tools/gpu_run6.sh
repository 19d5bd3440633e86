export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/sweep.py --tag r01e --rounds 3 > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"
head -45 gpurun_out/sweep.log | cut -c1-180
