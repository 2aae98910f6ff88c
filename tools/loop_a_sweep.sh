# the split of loop A between device and host, per contig group (HS_LOOP_A_DEVICE_MAX lists): step time and CPU of the default bench
mkdir -p gpurun_out/r6g
i=0
for T in 0 "1500,0" "1500,1500,0" "1500,1500,1500,0" "1500,1000,600,300,0" "3000,1500,800,0" "0"; do
  i=$((i+1))
  export HS_LOOP_A_DEVICE_MAX=$T
  timeout 300 python bench.py --steps 20 --warmup 3 --cpu-contigs 0 --no-f2f-job > gpurun_out/r6g/b_$i.json 2> gpurun_out/r6g/b_$i.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6g/b_$i.json").read().strip().split("\n")[-1])
k=d["kernels"]
print("T=$T ms/step %.2f cpu %.1f parity %s loop_a %s ship %.2f" % (d["ms_per_step"], d["host"]["process_cpu_ms_per_step"], d["parity"]["identical"], k.get("k_loop_a",{}).get("ms_per_step"), k["k_ship"]["ms_per_step"]))
PY
done
