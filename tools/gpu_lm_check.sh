cd /root/repo
python - <<'P'
import sys, os, subprocess, tempfile
sys.path.insert(0, "tests")
import golden_util as gu
import __graft_entry__ as ge
p = ge.paths()
td = tempfile.mkdtemp()
meta = gu.unpack("tetra25k_lowmem", td)
env = dict(os.environ, HS_TIMING="1", HS_SEED=str(meta.get("kwargs", {}).get("seed", 12345)))
r = subprocess.run([p["sr"], os.path.join(td, "variants.col"), "4", "0.05", os.path.join(td, "no_ploidy"), "1", "0.01", "0", os.path.join(td, "o.gro"), "0"], env=env, capture_output=True, text=True)
print(r.returncode)
print("\n".join(l for l in r.stderr.splitlines() if "low-memory" in l or "graph rows" in l))
P
