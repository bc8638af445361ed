"""Generate the V-HACD concave-movable fixtures (robovat_amd/assets/vhacd_hulls.json).

Runs ONLY in the build container: it executes the reference's prebuilt
`bin/vhacd` (V-HACD 2.3, the decomposition tool tools/convert_obj_to_urdf.py:19-32
drives) on procedurally authored concave OBJs (L, T, U, cross prisms) and keeps
the hull vertex lists (<= 16 vertices per hull, <= 4 hulls per body).  The
output JSON is data; the GPU box never needs the reference tree.

    python tests/golden/gen_vhacd_fixtures.py
"""
import json
import os
import re
import subprocess
import tempfile

import numpy as np
from scipy.spatial import ConvexHull

VHACD = '/root/reference/bin/vhacd'
OUT = os.path.join(os.path.dirname(__file__), '..', '..', 'robovat_amd', 'assets', 'vhacd_hulls.json')


def prism_obj(boxes):
    """Union of axis-aligned boxes as a triangle soup OBJ (V-HACD voxelises)."""
    verts, faces = [], []
    for (lo, hi) in boxes:
        base = len(verts)
        for x in (lo[0], hi[0]):
            for y in (lo[1], hi[1]):
                for z in (lo[2], hi[2]):
                    verts.append((x, y, z))
        quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
        for q in quads:
            faces.append((base + q[0], base + q[1], base + q[2]))
            faces.append((base + q[0], base + q[2], base + q[3]))
    lines = ['v %f %f %f' % v for v in verts] + ['f %d %d %d' % (a + 1, b + 1, c + 1) for a, b, c in faces]
    return '\n'.join(lines) + '\n'


T = 0.025  # bar half thickness (m)
H = 0.025  # half height
SHAPES = {
    'concave_L': [((-0.06, -0.06, -H), (-0.06 + 2 * T, 0.06, H)), ((-0.06, -0.06, -H), (0.06, -0.06 + 2 * T, H))],
    'concave_T': [((-0.07, 0.02, -H), (0.07, 0.02 + 2 * T, H)), ((-T, -0.07, -H), (T, 0.02, H))],
    'concave_U': [((-0.07, -0.05, -H), (-0.07 + 1.6 * T, 0.07, H)), ((0.07 - 1.6 * T, -0.05, -H), (0.07, 0.07, H)),
                  ((-0.07, -0.05, -H), (0.07, -0.05 + 1.6 * T, H))],
    'concave_X': [((-0.07, -T, -H), (0.07, T, H)), ((-T, -0.07, -H), (T, 0.07, H))],
}


def parse_wrl(text):
    hulls = []
    for m in re.finditer(r'point\s*\[(.*?)\]', text, re.S):
        nums = [float(x) for x in re.findall(r'[-+0-9.eE]+', m.group(1))]
        pts = np.array(nums).reshape(-1, 3)
        if len(pts) >= 4:
            hulls.append(pts)
    return hulls


def simplify(pts, max_verts=16):
    """Keep hull vertices; if more than max_verts, greedily drop the vertex whose
    removal loses the least volume."""
    pts = pts[ConvexHull(pts).vertices]
    while len(pts) > max_verts:
        vol = ConvexHull(pts).volume
        best, best_loss = None, None
        for i in range(len(pts)):
            q = np.delete(pts, i, axis=0)
            loss = vol - ConvexHull(q).volume
            if best is None or loss < best_loss:
                best, best_loss = i, loss
        pts = np.delete(pts, best, axis=0)
    return pts


def main():
    out = {}
    for name, boxes in SHAPES.items():
        with tempfile.TemporaryDirectory() as d:
            obj = os.path.join(d, 'in.obj'); wrl = os.path.join(d, 'out.wrl'); log = os.path.join(d, 'log.txt')
            with open(obj, 'w') as f:
                f.write(prism_obj(boxes))
            subprocess.run([VHACD, '--input', obj, '--output', wrl, '--log', log, '--maxhulls', '4',
                            '--maxNumVerticesPerCH', '16', '--resolution', '200000', '--concavity', '0.0002'],
                           check=True, stdout=subprocess.DEVNULL)
            hulls = parse_wrl(open(wrl).read())
        hulls = [simplify(h) for h in hulls][:4]
        out[name] = [np.round(h, 6).tolist() for h in hulls]
        print(name, 'hulls:', [len(h) for h in hulls])
    with open(OUT, 'w') as f:
        json.dump(out, f)
    print('wrote', os.path.abspath(OUT))


if __name__ == '__main__':
    main()
