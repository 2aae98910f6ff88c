# alternating runs of the default bench with and without an environment switch on one box: tools/ab_env.sh <VAR=value> [runs] [extra bench args...]
sw="$1"; runs="${2:-3}"; shift; shift
mkdir -p gpurun_out/ab
for i in $(seq 1 $runs); do
  for mode in off on; do
    if [ $mode = on ]; then export "$sw"; else unset "${sw%%=*}"; fi
    timeout 300 python bench.py --steps 20 --warmup 3 --cpu-contigs 0 --no-f2f-job "$@" > gpurun_out/ab/b_${mode}_$i.json 2> gpurun_out/ab/b_${mode}_$i.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/b_${mode}_$i.json").read().strip().split("\n")[-1])
print("$sw $mode run $i: ms/step %.2f cpu %.1f parity %s" % (d["ms_per_step"], d["host"]["process_cpu_ms_per_step"], d["parity"]["identical"]))
PY
  done
done
