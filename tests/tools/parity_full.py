"""Parity at BASELINE's full configurations: the drop-in executables against the compiled reference (oracle/_ref, with the
pinned random_device for stage 4) on the same synthetic files; per-contig comparison of .col / .vcf / error_rate / .gro, and the .gaf of the next stage.
Usage: python tests/tools/parity_full.py C3|C4|C5 [n_contigs]       (prints one JSON line; needs a GPU and oracle/_ref)"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from hairsplitter_amd import synth, canon
    import bench
    import __graft_entry__ as ge
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    count = int(sys.argv[2]) if len(sys.argv) > 2 else None
    p = ge.paths()
    threads = bench.effective_cores()
    t = time.perf_counter()
    contigs = synth.config_contigs(cfg, count=count)
    t_gen = time.perf_counter() - t
    bp = sum(c.aligned_bp for c in contigs)
    out = {"config": cfg, "contigs": len(contigs), "aligned_bp": bp, "threads": threads, "generation_s": round(t_gen, 1)}
    with tempfile.TemporaryDirectory() as td:
        f = synth.write_files(contigs, td)
        res = {}
        for tag, cv, sr in (("hip", p["cv"], p["sr"]), ("ref", p["ref_cv"], p["ref_sr_seeded"])):
            col, vcf, err, gro = (os.path.join(td, tag + x) for x in (".col", ".vcf", ".err", ".gro"))
            t0 = time.perf_counter()
            subprocess.run([cv, f["gfa"], f["reads"], f["sam"], str(threads), td, err, "0", "0", col, vcf, "0.33"], check=True, stdout=subprocess.DEVNULL)
            t1 = time.perf_counter()
            e = bench.py_error_rate(float(open(err).read().strip()))
            subprocess.run([sr, col, str(threads), str(e), os.path.join(td, "no_ploidy"), "0", "0.01", "0", gro, "0"], check=True, stdout=subprocess.DEVNULL)
            t2 = time.perf_counter()
            res[tag] = (col, vcf, err, gro)
            out[tag] = {"call_variants_s": round(t1 - t0, 2), "separate_reads_s": round(t2 - t1, 2)}
        a, b = res["hip"], res["ref"]
        out["col_identical"] = canon.split_blocks(a[0]) == canon.split_blocks(b[0])
        out["vcf_identical"] = canon.vcf_blocks(a[1]) == canon.vcf_blocks(b[1])
        out["error_rate_identical"] = open(a[2]).read() == open(b[2]).read()
        ga, gb = canon.split_blocks(a[3]), canon.split_blocks(b[3])
        out["gro_identical"] = ga == gb
        if ga != gb:
            out["gro_diff"] = canon.diff_blocks(ga, gb)[:3]
        out["n_snps"] = sum(1 for l in open(a[0]) if l.startswith("SNPS"))
        out["n_groups"] = sum(1 for l in open(a[3]) if l.startswith("GROUP"))
        # next stage: the .gaf derived from each side's own .gro (hs_gro_to_gaf vs the reference's HS_create_new_contigs, which
        # writes the .gaf and then stops at its first external tool)
        if os.path.exists(p.get("ref_cnc", "")):
            gaf_a, gaf_b, tmp = os.path.join(td, "hip.gaf"), os.path.join(td, "ref.gaf"), os.path.join(td, "cnc_tmp")
            os.makedirs(tmp, exist_ok=True)
            t0 = time.perf_counter()
            subprocess.run([p["gaf"], f["gfa"], f["reads"], f["sam"], a[3], "0", gaf_a, str(threads)], check=True, stdout=subprocess.DEVNULL)
            t1 = time.perf_counter()
            subprocess.run([p["ref_cnc"], f["gfa"], f["reads"], "0.05", b[3], f["sam"], tmp + "/", str(threads), "ont", os.path.join(tmp, "o.gfa"), gaf_b,
                            "racon", "0", "0", "/nonexistent/minimap2", "/nonexistent/racon", "/nonexistent/medaka", "/nonexistent/samtools",
                            "/nonexistent/python", "0"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            t2 = time.perf_counter()
            out["gaf"] = {"hip_s": round(t1 - t0, 2), "ref_until_first_external_tool_s": round(t2 - t1, 2),
                          "identical": os.path.exists(gaf_b) and open(gaf_a, "rb").read() == open(gaf_b, "rb").read(),
                          "lines": sum(1 for _ in open(gaf_a))}
        out["speedup_file_to_file"] = round((out["ref"]["call_variants_s"] + out["ref"]["separate_reads_s"]) / (out["hip"]["call_variants_s"] + out["hip"]["separate_reads_s"]), 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
