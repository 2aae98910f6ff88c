"""The product's host side (file boundary, loops A/B, window planning, cluster merging; device interface served by the oracle)
under AddressSanitizer + UBSan: every golden case through tests/harness/_build/host_harness_asan, outputs compared with the
goldens. CPU only (sanitizers do not run on the GPU pool).  usage: python tools/host_asan.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "harness"), "asan"], check=True)
    h = os.path.join(ROOT, "tests", "harness", "_build", "host_harness_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    bad = 0
    for case in gu.case_names():
        with tempfile.TemporaryDirectory() as td:
            meta = gu.unpack(case, td)
            try:
                diff = gu.compare(td, gu.run_stage_pair([h, "call_variants"], [h, "separate_reads"], td, meta, env=env))
            except Exception as e:      # a sanitizer report ends the process with a non-zero status
                diff = [repr(e)[:400]]
            print(case, "OK" if not diff else "FAILED %r" % diff[:2], flush=True)
            bad += bool(diff)
    print("cases with a finding:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
