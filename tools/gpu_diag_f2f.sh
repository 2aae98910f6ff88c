# why are the file-to-file timings of tests/test_gpu_full_configs.py slower after the drop-in tests?
cd /root/repo
mkdir -p gpurun_out
L=gpurun_out/diag_f2f.log
: > $L
echo "== C2 alone, fresh box" >> $L
HS_TIMING=1 timeout 600 python -m pytest tests/test_gpu_full_configs.py -q -m gpu -k c2_x16 -s >> $L 2>&1
tail -1 gpurun_out/parity_full_configs.jsonl | cut -c1-300 >> $L
echo "== drop-in tests" >> $L
timeout 900 python -m pytest tests/test_gpu_dropin.py -q -m gpu -x 2>&1 | tail -2 >> $L
echo "== processes left" >> $L
ps -eo pid,ppid,stat,etime,pcpu,rss,comm | grep -v "ps\|grep\|bash\|sleep" | head -40 >> $L
echo "== C2 after the drop-in tests" >> $L
HS_TIMING=1 timeout 600 python -m pytest tests/test_gpu_full_configs.py -q -m gpu -k c2_x16 -s >> $L 2>&1
tail -1 gpurun_out/parity_full_configs.jsonl | cut -c1-300 >> $L
ps -eo pid,ppid,stat,etime,pcpu,rss,comm | grep -v "ps\|grep\|bash\|sleep" | head -40 >> $L
grep -v "^\[hs timing\] stamp" $L | tail -60
