# the default bench command N times, each bounded (stack dump on a stall): does every run come back?
cd /root/repo
for i in $(seq 1 ${1:-4}); do echo "== run $i"; timeout -s USR1 -k 15 240 python bench.py 2>gpurun_out/repeat_$i.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2), 'G bp/s', round(d['ms_per_step'],2), 'ms/step, f2f job', round(d['file_to_file']['job']['dropin_s']['total'],2), 's')" || tail -30 gpurun_out/repeat_$i.err; done
