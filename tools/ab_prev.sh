set -e
cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/ab_parity.log 2>&1 || { tail -30 gpurun_out/ab_parity.log; exit 1; }
tail -2 gpurun_out/ab_parity.log
for i in 1 2; do
RT_LIB_OVERRIDE=accelerated-ray-tracer_amd/lib/ab/librt_prev.so python tools/sweep.py --ns 500 --rounds 4 "" | tee -a gpurun_out/ab_prev.log
python tools/sweep.py --ns 500 --rounds 4 "" | tee -a gpurun_out/ab_new.log
done
RT_LIB_OVERRIDE=accelerated-ray-tracer_amd/lib/ab/librt_prev.so python tools/sweep.py --scene cornell --nx 600 --ny 600 --ns 1000 --rounds 2 "" | tee -a gpurun_out/ab_prev.log
python tools/sweep.py --scene cornell --nx 600 --ny 600 --ns 1000 --rounds 2 "" | tee -a gpurun_out/ab_new.log
RT_LIB_OVERRIDE=accelerated-ray-tracer_amd/lib/ab/librt_prev.so python tools/sweep.py --scene final --nx 800 --ny 800 --ns 200 --rounds 2 "" | tee -a gpurun_out/ab_prev.log
python tools/sweep.py --scene final --nx 800 --ny 800 --ns 200 --rounds 2 "" | tee -a gpurun_out/ab_new.log
