"""Parity at BASELINE's full configurations (see tests/full_configs.py; the same checks run as -m gpu tests in
tests/test_gpu_full_configs.py). Usage: python tests/tools/parity_full.py C2|C3|C4|C5|C5U [n_contigs]   (prints one JSON line; needs a GPU and oracle/_ref)"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import full_configs as fc
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    count = int(sys.argv[2]) if len(sys.argv) > 2 else None
    with tempfile.TemporaryDirectory() as td:
        print(json.dumps(fc.run_config(cfg, td, count)))


if __name__ == "__main__":
    main()
