"""Write the gfx950 code object inside librovat_hip.so to a file: python tools/extract_co.py [lib.so] out.co"""
import os, struct, sys
path = sys.argv[1] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), '..', 'robovat_amd', 'librovat_hip.so')
out = sys.argv[-1]
d = open(path, 'rb').read()
i = d.find(b'__CLANG_OFFLOAD_BUNDLE__')
n = struct.unpack_from('<Q', d, i + 24)[0]
off = i + 32
for _ in range(n):
    o, sz, ts = struct.unpack_from('<QQQ', d, off); off += 24
    t = d[off:off + ts].decode(); off += ts
    if 'gfx950' in t:
        open(out, 'wb').write(d[i + o:i + o + sz])
        print('wrote', out, sz, 'bytes', t)
