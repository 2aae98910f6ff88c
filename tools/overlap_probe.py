"""What would overlapping the streaming pass with the contig groups buy? (diagnostic for the next round, not a measurement:
the error rate handed to stage 4 is a constant here, so that the second half's streaming pass may run while the first
half's groups are already at work.) Usage: python tools/overlap_probe.py [contigs=256] [parts=2] [steps=12]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    import bench
    from hairsplitter_amd import synth
    contigs = synth.generate_job("C2", list(range(n)), seed=2, workers=8)[0]
    import torch
    from hairsplitter_amd import api
    torch.cuda.set_device(0)
    api.require_gpu()
    threads = 64
    whole = api.PipelineGroups(contigs, 8)
    halves = [api.PipelineGroups(contigs[i * n // parts:(i + 1) * n // parts], max(1, 8 // parts)) for i in range(parts)]
    fixed = lambda cv: 0.0543   # noqa: E731
    bp = whole.aligned_bp

    def step_whole():
        whole.run(0.33, threads, fixed, rarest_strain_abundance=0.01, window_size=2000)

    def step_parts(stagger_ms):
        ths = []
        for i, h in enumerate(halves):
            t = threading.Thread(target=lambda h=h: h.run(0.33, threads // parts, fixed, rarest_strain_abundance=0.01, window_size=2000))
            t.start(); ths.append(t)
            if i + 1 < len(halves):
                time.sleep(stagger_ms * 1e-3)
        for t in ths:
            t.join()

    for name, fn in (("one batch, streaming pass then groups", step_whole),) + tuple(
            (f"{parts} sub-batches, second started {s} ms after the first", (lambda s=s: step_parts(s))) for s in (0.0, 1.5, 2.5, 4.0)):
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"{name}: {dt * 1e3:.2f} ms/step = {bp / dt / 1e9:.1f} G aligned bp/s", flush=True)


if __name__ == "__main__":
    main()
