"""Registers / LDS / scratch of the kernels in a built librovat_hip.so (the gfx950 code object inside the
clang offload bundle): python tools/kernel_resources.py [lib.so]"""
import re, struct, subprocess, sys, tempfile, os
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), '..', 'robovat_amd', 'librovat_hip.so')
d = open(path, 'rb').read()
# one offload bundle per translation unit (rv_kernels.hip, rv_kernels_occ2.hip): read the gfx950 code object of each
txt = ''
start = 0
while True:
    i = d.find(b'__CLANG_OFFLOAD_BUNDLE__', start)
    if i < 0:
        break
    start = i + 24
    n = struct.unpack_from('<Q', d, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, ts = struct.unpack_from('<QQQ', d, off); off += 24
        t = d[off:off + ts].decode(); off += ts
        if 'gfx950' in t:
            with tempfile.NamedTemporaryFile(suffix='.o', delete=False) as f:
                f.write(d[i + o:i + o + sz]); co = f.name
            txt += subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
            os.unlink(co)
rows = re.findall(r'\.agpr_count:\s*(\d+).*?\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)', txt, re.S)
print('%-44s %6s %6s %8s %9s %6s %6s' % ('kernel', 'vgpr', 'agpr', 'LDS B', 'scratch B', 'sgpr', 'spill'))
for a, lds, name, scr, sg, vg, sp in rows:
    print('%-44s %6s %6s %8s %9s %6s %6s' % (name[:44], vg, a, lds, scr, sg, sp))
