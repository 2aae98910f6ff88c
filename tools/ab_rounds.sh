# the round-5 tree (worktree _r05 at a5ba4de, built in place) against this tree on ONE box, alternating runs of the default job
for i in 1 2 3 4; do
  for t in r05 r06; do
    if [ $t = r05 ]; then d=_r05; else d=.; fi
    ( cd $d && timeout 300 python bench.py --steps 30 --warmup 3 --cpu-contigs 0 --no-f2f-job > /tmp/ab_$t.json 2>/tmp/ab_$t.err )
    python - <<PY
import json
d=json.loads(open("/tmp/ab_$t.json").read().strip().split("\n")[-1])
print("$t run $i: ms/step %.2f cpu %.1f parity %s kernels %.2f ms" % (d["ms_per_step"], d["host"]["process_cpu_ms_per_step"], (d.get("parity") or {}).get("identical"), sum(v["ms_per_step"] for v in d["kernels"].values())))
PY
  done
done
