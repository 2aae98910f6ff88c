"""Malformed inputs through the ASan/UBSan host harness (make -C tests/harness asan): seeded random damage to the GFA, the reads
file or the SAM of four golden cases; the file boundary (hs_io.cpp) and what follows must end with an exit status, never with a
sanitizer report or a signal. Stage 4 then runs on whatever stage 3 wrote.  usage: python tools/host_fuzz_inputs.py [seed] [cases per golden]
(the .col parser was fuzzed the same way: it is how the read-index check of hs::parse_col came about)"""
import os, sys, random, shutil, subprocess, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import golden_util as gu
H = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'harness', '_build', 'host_harness_asan')      # make -C tests/harness asan
env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
def mutate(path):
    s = bytearray(open(path, 'rb').read())
    if not s: return
    for _ in range(rnd.randint(1, 5)):
        k = rnd.randrange(len(s)); op = rnd.randint(0, 5)
        if op == 0: s[k] = 9
        elif op == 1: s[k] = 10
        elif op == 2: del s[k:k + rnd.randint(1, 60)]
        elif op == 3: s[k] = rnd.choice(b'SLMIDX=*0123456789-+@>\x00\xff')
        elif op == 4: s[k:k] = bytes(rnd.choice(b'0123456789MIDSH') for _ in range(rnd.randint(1, 12)))
        else: del s[k:]            # truncated file
    open(path, 'wb').write(bytes(s))
findings = 0; rcs = {}
for case in ('simple_mock', 'edge_ops', 'clips', 'dip10k_fastq'):
    if case not in gu.case_names(): continue
    for it in range(N):
        with tempfile.TemporaryDirectory() as td:
            meta = gu.unpack(case, td)
            which = rnd.choice(['assembly.gfa', 'aln.sam', 'reads'])
            path = gu.reads_path(td, meta) if which == 'reads' else os.path.join(td, which)
            mutate(path)
            kw = meta.get('kwargs', {})
            col, vcf, err, gro = (os.path.join(td, 'f_' + n) for n in ('variants.col', 'variants.vcf', 'error_rate.txt', 'reads_haplo.gro'))
            r = subprocess.run([H, 'call_variants', os.path.join(td, 'assembly.gfa'), gu.reads_path(td, meta), os.path.join(td, 'aln.sam'), '1', td, err,
                                str(kw.get('amplicon', 0)), '0', col, vcf, '0.33'], capture_output=True, env=env, timeout=300)
            rcs[r.returncode] = rcs.get(r.returncode, 0) + 1
            out = r.stdout + r.stderr
            if r.returncode < 0 or r.returncode == 134 or b'Sanitizer' in out or b'runtime error' in out:
                findings += 1
                print('FINDING', case, which, r.returncode, out[-1500:].decode(errors='replace'), flush=True)
                shutil.copy(path, os.path.join(tempfile.gettempdir(), 'fuzz_%s_%d_%s' % (case, it, os.path.basename(path))))
            elif r.returncode == 0 and os.path.exists(col):
                # stage 4 on whatever stage 3 wrote
                r2 = subprocess.run([H, 'separate_reads', col, '1', meta['error_rate_arg'], os.path.join(td, 'absent_ploidy.txt'), '0', '0.01', str(kw.get('amplicon', 0)), gro, '0'],
                                    capture_output=True, env=env, timeout=300)
                out2 = r2.stdout + r2.stderr
                if r2.returncode < 0 or r2.returncode == 134 or b'Sanitizer' in out2 or b'runtime error' in out2:
                    findings += 1
                    print('FINDING stage 4', case, which, r2.returncode, out2[-1500:].decode(errors='replace'), flush=True)
print('exit codes of stage 3:', rcs, 'findings:', findings)
