# VGPRs / SGPRs / scratch / LDS of every kernel of the library, from the gfx950 assembly (no GPU needed)
cd "$(dirname "$0")/../hairsplitter_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fgpu-default-stream=per-thread \
    --cuda-device-only -S -o /tmp/hs_capi.s hs_capi.hip 2>/dev/null
python3 - <<'P'
import re
txt = open('/tmp/hs_capi.s').read()
print("%-44s %5s %5s %8s %7s" % ("kernel", "vgpr", "sgpr", "scratch", "lds"))
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', body) or [None, '?'])[1]
    nm = re.sub(r'^_ZN5hsdev\d*', '', name)
    nm = re.sub(r'E[PKvilhjmxyab].*$', '', nm)[:44]
    print("%-44s %5s %5s %8s %7s" % (nm, g('next_free_vgpr'), g('next_free_sgpr'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
P
