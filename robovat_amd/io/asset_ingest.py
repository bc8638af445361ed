"""Asset ingest: RoboVat's movable assets -> the shape templates of ``rv_scene``.

The reference loads a movable with ``BulletPhysics.add_body(filename=<urdf>, pose, scale)``
(``bullet_physics.py:143-186``); its movable URDFs are the ones ``tools/convert_obj_to_urdf.py``
writes from ``tools/templates/urdf_template.xml``: ONE link with a ``<contact>`` block
(lateral / rolling / spinning friction), an ``<inertial>`` block and one ``<collision>`` mesh per
V-HACD part (``collision_template.xml``: an OBJ file and a uniform scale).  This module reads such a
URDF (boxes / cylinders / spheres as collision geometry are accepted too), turns every collision
part into a convex hull with at most ``RV_MAXV`` vertices, merges parts down to ``RV_MAXH`` hulls and
returns what ``scenes.make_scene(shape_hulls=...)`` takes.  ``arm_chain_from_urdf`` reads the serial
chain of a robot URDF (joint origins, axes, limits) into the arrays ``rv_arm`` is filled from.

Nothing here runs on the hot path: it is the host-side door through which real ``assets/`` enter once
they are available (they are not distributed with the reference).
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from robovat_amd import abi


# ------------------------------------------------------------------------ OBJ
def read_obj(path):
    """Vertices [n, 3] and triangles [m, 3] (0-based; polygons are fanned) of a Wavefront OBJ."""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == 'v' and len(t) >= 4:
                verts.append([float(t[1]), float(t[2]), float(t[3])])
            elif t[0] == 'f' and len(t) >= 4:
                idx = []
                for tok in t[1:]:
                    i = int(tok.split('/')[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    if not verts:
        raise ValueError('no vertices in %s' % path)
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64).reshape(-1, 3)


# ---------------------------------------------------------------- convex hulls
def _hull_volume(pts):
    from scipy.spatial import ConvexHull
    return float(ConvexHull(pts).volume)


def convex_hull_reduced(points, max_verts=abi.RV_MAXV):
    """Vertices of the convex hull of ``points``, thinned to at most ``max_verts`` by dropping, one at a
    time, the hull vertex whose removal loses the least volume (the hull only ever shrinks, so the
    template never collides where the mesh does not)."""
    from scipy.spatial import ConvexHull
    pts = np.asarray(points, dtype=np.float64)
    if len(pts) < 4:
        raise ValueError('a collision part needs at least 4 points')
    span = np.ptp(pts, axis=0)
    if span.min() <= 1e-9 * max(span.max(), 1e-12):
        raise ValueError('degenerate (flat) collision part')
    try:
        v = pts[ConvexHull(pts).vertices]
    except Exception as ex:       # scipy's QhullError: a flat part that is not axis-aligned
        raise ValueError('degenerate (flat) collision part: %s' % str(ex).splitlines()[0])
    while len(v) > max_verts:
        if len(v) > 4 * max_verts:
            # far too many: coarse pre-thinning by farthest-point sampling
            keep = [int(np.argmax(np.linalg.norm(v - v.mean(0), axis=1)))]
            d = np.linalg.norm(v - v[keep[0]], axis=1)
            while len(keep) < 2 * max_verts:
                k = int(np.argmax(d)); keep.append(k)
                d = np.minimum(d, np.linalg.norm(v - v[k], axis=1))
            v = v[keep]
            v = v[ConvexHull(v).vertices]
            continue
        full = _hull_volume(v)
        loss = []
        for i in range(len(v)):
            rest = np.delete(v, i, axis=0)
            try:
                loss.append(full - _hull_volume(rest))
            except Exception:           # removal would flatten the hull
                loss.append(np.inf)
        v = np.delete(v, int(np.argmin(loss)), axis=0)
        v = v[ConvexHull(v).vertices]
    return v


def merge_hulls(hulls, max_hulls=abi.RV_MAXH, max_verts=abi.RV_MAXV, max_added_fraction=0.25):
    """At most ``max_hulls`` hulls: the pair whose union hull adds the least volume is merged first.

    Merging FILLS the concavity between two parts (the template then collides where the mesh does
    not), unlike the vertex thinning of ``convex_hull_reduced``, which only ever shrinks a hull.  The
    volume added by all merges is therefore measured: above ``max_added_fraction`` of the volume of
    the parts a ``ValueError`` asks for fewer / coarser V-HACD parts (or a larger RV_MAXH), below it
    a ``UserWarning`` reports the figure."""
    import warnings
    hulls = [np.asarray(h, dtype=np.float64) for h in hulls]
    parts_volume = sum(_hull_volume(h) for h in hulls)
    added = 0.0
    n_in = len(hulls)
    while len(hulls) > max_hulls:
        best = None
        vol = [_hull_volume(h) for h in hulls]
        for i in range(len(hulls)):
            for j in range(i + 1, len(hulls)):
                u = _hull_volume(np.concatenate([hulls[i], hulls[j]]))
                extra = u - vol[i] - vol[j]
                if best is None or extra < best[0]:
                    best = (extra, i, j)
        extra, i, j = best
        added += max(extra, 0.0)
        merged = convex_hull_reduced(np.concatenate([hulls[i], hulls[j]]), max_verts)
        hulls = [h for k, h in enumerate(hulls) if k not in (i, j)] + [merged]
    if n_in > max_hulls:
        frac = added / max(parts_volume, 1e-30)
        msg = ('%d collision parts merged into %d hulls: the merges fill %.1f %% of the parts\' volume '
               '(concavities the mesh has and the template does not)' % (n_in, max_hulls, 100.0 * frac))
        if frac > max_added_fraction:
            raise ValueError(msg + '; above the %.0f %% limit -- decompose into at most %d parts' % (100.0 * max_added_fraction, max_hulls))
        warnings.warn(msg)
    return hulls


# ------------------------------------------------------------------------ URDF
def _floats(s, n=None, default=None):
    if s is None:
        return default
    v = [float(x) for x in s.split()]
    if n is not None and len(v) != n:
        raise ValueError('expected %d numbers, got %r' % (n, s))
    return v


def _rpy_matrix(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _geometry_points(geom, base_dir):
    """Points whose convex hull is the collision geometry (in the geometry's own frame)."""
    mesh = geom.find('mesh')
    if mesh is not None:
        fn = mesh.get('filename')
        if fn.startswith('package://'):
            fn = fn[len('package://'):]
        path = fn if os.path.isabs(fn) else os.path.join(base_dir, fn)
        if not path.lower().endswith('.obj'):
            raise ValueError('only OBJ collision meshes are supported: %s' % fn)   # bullet_physics.py:184
        v, _ = read_obj(path)
        return v * np.asarray(_floats(mesh.get('scale'), 3, [1.0, 1.0, 1.0]))
    box = geom.find('box')
    if box is not None:
        h = 0.5 * np.asarray(_floats(box.get('size'), 3))
        return np.array([[sx * h[0], sy * h[1], sz * h[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    cyl = geom.find('cylinder')
    if cyl is not None:
        r, hl = float(cyl.get('radius')), 0.5 * float(cyl.get('length'))
        a = np.arange(8) * 2 * np.pi / 8
        ring = np.stack([r * np.cos(a), r * np.sin(a)], axis=1)
        return np.concatenate([np.c_[ring, np.full(8, hl)], np.c_[ring, np.full(8, -hl)]])
    sph = geom.find('sphere')
    if sph is not None:
        r = float(sph.get('radius'))
        g = (1 + 5 ** 0.5) / 2
        ico = np.array([[0, 1, g], [0, -1, g], [0, 1, -g], [0, -1, -g], [1, g, 0], [-1, g, 0], [1, -g, 0], [-1, -g, 0],
                        [g, 0, 1], [-g, 0, 1], [g, 0, -1], [-g, 0, -1]], dtype=np.float64)
        return r * ico / np.linalg.norm(ico[0])
    raise ValueError('unsupported collision geometry')


def read_movable_urdf(path, scale=1.0):
    """One-link movable (the URDFs of ``convert_obj_to_urdf.py``): dict with ``name``, ``hulls`` (list of
    [n, 3] arrays, link frame, x ``scale``), ``mass``, ``com``, ``inertia`` (3x3, as authored), and the
    ``<contact>`` coefficients ``lateral_friction`` / ``rolling_friction`` / ``spinning_friction``."""
    if not os.path.exists(path):
        raise ValueError('The path %s does not exist.' % path)           # bullet_physics.py:162
    root = ET.parse(path).getroot()
    links = root.findall('link')
    if len(links) != 1:
        raise ValueError('a movable has exactly one link, %s has %d' % (path, len(links)))
    link = links[0]
    base_dir = os.path.dirname(os.path.abspath(path))
    parts = []
    for col in link.findall('collision'):
        org = col.find('origin')
        xyz = np.asarray(_floats(org.get('xyz') if org is not None else None, 3, [0.0, 0.0, 0.0]))
        rpy = _floats(org.get('rpy') if org is not None else None, 3, [0.0, 0.0, 0.0])
        pts = _geometry_points(col.find('geometry'), base_dir)
        parts.append((pts @ _rpy_matrix(*rpy).T + xyz) * float(scale))
    if not parts:
        raise ValueError('%s has no collision geometry' % path)
    hulls = merge_hulls([convex_hull_reduced(p) for p in parts])
    out = {'name': root.get('name', os.path.splitext(os.path.basename(path))[0]), 'hulls': hulls,
           'mass': None, 'com': np.zeros(3), 'inertia': None,
           'lateral_friction': 1.0, 'rolling_friction': 0.0, 'spinning_friction': 0.0}
    contact = link.find('contact')
    if contact is not None:
        for key in ('lateral_friction', 'rolling_friction', 'spinning_friction'):
            el = contact.find(key)
            if el is not None:
                out[key] = float(el.get('value'))
    inertial = link.find('inertial')
    if inertial is not None:
        m = inertial.find('mass')
        if m is not None:
            out['mass'] = float(m.get('value'))
        org = inertial.find('origin')
        if org is not None:
            out['com'] = np.asarray(_floats(org.get('xyz'), 3, [0.0, 0.0, 0.0])) * float(scale)
        it = inertial.find('inertia')
        if it is not None:
            g = {k: float(it.get(k, 0.0)) for k in ('ixx', 'ixy', 'ixz', 'iyy', 'iyz', 'izz')}
            out['inertia'] = np.array([[g['ixx'], g['ixy'], g['ixz']], [g['ixy'], g['iyy'], g['iyz']], [g['ixz'], g['iyz'], g['izz']]])
    return out


def shape_library_from_urdfs(paths, scale=1.0):
    """``[(name, hulls), ...]`` for ``scenes.make_scene(shape_hulls=...)`` plus the per-asset contact /
    mass data (the env config's MASS / FRICTION ranges override them in PushEnv, push_env.py:446-462)."""
    lib, meta = [], {}
    for p in paths:
        m = read_movable_urdf(p, scale)
        lib.append((m['name'], m['hulls']))
        meta[m['name']] = {k: m[k] for k in ('mass', 'lateral_friction', 'rolling_friction', 'spinning_friction')}
    if len(lib) > abi.RV_MAX_SHAPES:
        raise ValueError('at most %d shape templates per scene' % abi.RV_MAX_SHAPES)
    return lib, meta


# ------------------------------------------------------------------- arm chain
def _axis_to_z_quat(axis):
    """xyzw quaternion that rotates z onto ``axis``."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    z = np.array([0.0, 0.0, 1.0])
    c = float(np.dot(z, a))
    if c > 1.0 - 1e-12:
        return np.array([0.0, 0.0, 0.0, 1.0])
    if c < -1.0 + 1e-12:
        return np.array([1.0, 0.0, 0.0, 0.0])
    v = np.cross(z, a)
    q = np.array([v[0], v[1], v[2], 1.0 + c])
    return q / np.linalg.norm(q)


def _quat_from_matrix(R):
    from robovat_amd.math import rotations
    return np.asarray(rotations.quaternion_from_matrix3(np.asarray(R, dtype=np.float64)), dtype=np.float64)


def _quat_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def arm_chain_from_urdf(path, tip_link, base_link=None):
    """The serial chain base -> ``tip_link`` of a robot URDF as the arrays ``rv_arm`` holds:
    per MOVING joint i its origin in the parent joint frame (``jpos``), the fixed rotation into a frame
    whose z axis is the joint axis (``jquat``, xyzw: the kernel's joints turn about local z), limits
    (``q_lo, q_hi, v_max, effort``) and names; fixed joints are folded into the next origin.  The last
    entry (``tip``) is the fixed transform from the last moving joint's frame to the tip link."""
    root = ET.parse(path).getroot()
    joints = {j.find('child').get('link'): j for j in root.findall('joint')}
    chain, link = [], tip_link
    while link in joints and link != base_link:
        j = joints[link]
        chain.append(j)
        link = j.find('parent').get('link')
    if base_link is not None and link != base_link:
        raise ValueError('%s is not an ancestor of %s' % (base_link, tip_link))
    chain.reverse()
    out = {'names': [], 'jpos': [], 'jquat': [], 'q_lo': [], 'q_hi': [], 'v_max': [], 'effort': [], 'kind': [],
           'colliders': [],                    # per moving joint: [n, 3] collision points of the links riding on it, kernel frame
           'inertials': []}                    # per moving joint: [(mass, com[3], inertia 3x3 about the com)] of those links, kernel frame
    links = {l.get('name'): l for l in root.findall('link')}
    base_dir = os.path.dirname(os.path.abspath(path))

    def link_points(name):
        pts = []
        for col in links[name].findall('collision') if name in links else []:
            geom = col.find('geometry')
            if geom is None:
                continue
            org = col.find('origin')
            xyz = np.asarray(_floats(org.get('xyz') if org is not None else None, 3, [0.0, 0.0, 0.0]))
            R = _rpy_matrix(*_floats(org.get('rpy') if org is not None else None, 3, [0.0, 0.0, 0.0]))
            g = np.asarray(_geometry_points(geom, base_dir), dtype=np.float64)
            sph, cyl = geom.find('sphere'), geom.find('cylinder')
            if sph is not None:                 # (the box of a round shape: its axis extremes, not the polyhedron's)
                r = float(sph.get('radius')); g = np.concatenate([g, r * np.concatenate([np.eye(3), -np.eye(3)])])
            if cyl is not None:
                r = float(cyl.get('radius')); g = np.concatenate([g, r * np.array([[1.0, 1, 0], [-1, -1, 0]])])
            pts.append(g @ R.T + xyz)
        return np.concatenate(pts) if pts else np.zeros((0, 3))
    def link_inertial(name):
        ine = links[name].find('inertial') if name in links else None
        if ine is None or ine.find('mass') is None:
            return None
        org = ine.find('origin')
        xyz = np.asarray(_floats(org.get('xyz') if org is not None else None, 3, [0.0, 0.0, 0.0]))
        R = _rpy_matrix(*_floats(org.get('rpy') if org is not None else None, 3, [0.0, 0.0, 0.0]))
        it = ine.find('inertia')
        g = (lambda k: float(it.get(k, 0.0))) if it is not None else (lambda k: 0.0)
        I = np.array([[g('ixx'), g('ixy'), g('ixz')], [g('ixy'), g('iyy'), g('iyz')], [g('ixz'), g('iyz'), g('izz')]])
        return float(ine.find('mass').get('value')), xyz, R @ I @ R.T
    T_R, T_p = np.eye(3), np.zeros(3)          # pending fixed transform, in the frame of the last moving joint
    A_prev = np.eye(3)                         # rotation URDF-joint-frame -> kernel-joint-frame of the previous moving joint
    for j in chain:
        org = j.find('origin')
        xyz = np.asarray(_floats(org.get('xyz') if org is not None else None, 3, [0.0, 0.0, 0.0]))
        rpy = _floats(org.get('rpy') if org is not None else None, 3, [0.0, 0.0, 0.0])
        R = _rpy_matrix(*rpy)
        p_new, R_new = T_p + T_R @ xyz, T_R @ R
        kind = j.get('type')
        if kind == 'fixed':
            T_R, T_p = R_new, p_new
            # a link behind a fixed joint rides on the last moving joint: its collision points in that frame
            if out['colliders']:
                lp = link_points(j.find('child').get('link'))
                if len(lp):
                    out['colliders'][-1] = np.concatenate([out['colliders'][-1], (lp @ T_R.T + T_p) @ A_prev])
                li = link_inertial(j.find('child').get('link'))
                if li is not None:
                    Rk = A_prev.T @ T_R
                    out['inertials'][-1].append((li[0], A_prev.T @ (T_R @ li[1] + T_p), Rk @ li[2] @ Rk.T))
            continue
        if kind not in ('revolute', 'continuous', 'prismatic'):
            raise ValueError('unsupported joint type %s' % kind)
        ax = j.find('axis')
        axis = np.asarray(_floats(ax.get('xyz') if ax is not None else None, 3, [1.0, 0.0, 0.0]))
        Az = _quat_matrix(_axis_to_z_quat(axis))           # kernel frame = URDF joint frame . Az
        # in the previous kernel frame: position A_prev^T p, rotation A_prev^T R_new Az
        out['jpos'].append((A_prev.T @ p_new).tolist())
        out['jquat'].append(_quat_from_matrix(A_prev.T @ R_new @ Az).tolist())
        lim = j.find('limit')
        lo = float(lim.get('lower', -np.pi)) if lim is not None and kind != 'continuous' else -np.pi
        hi = float(lim.get('upper', np.pi)) if lim is not None and kind != 'continuous' else np.pi
        out['q_lo'].append(lo); out['q_hi'].append(hi)
        out['v_max'].append(float(lim.get('velocity', 1.0)) if lim is not None else 1.0)
        out['effort'].append(float(lim.get('effort', 0.0)) if lim is not None else 0.0)
        out['names'].append(j.get('name')); out['kind'].append(kind)
        T_R, T_p, A_prev = np.eye(3), np.zeros(3), Az
        out['colliders'].append(link_points(j.find('child').get('link')) @ Az)     # (row vectors: p_kernel = Az^T p_urdf)
        li = link_inertial(j.find('child').get('link'))
        out['inertials'].append([(li[0], Az.T @ li[1], Az.T @ li[2] @ Az)] if li is not None else [])
    out['tip'] = {'pos': (A_prev.T @ T_p).tolist(), 'quat': _quat_from_matrix(A_prev.T @ T_R).tolist()}
    return out


def arm_from_urdf(path, tip_link, base_link=None, base_pos=(0.0, 0.0, 0.0), base_rpy=(0.0, 0.0, 0.0), max_accel=None,
                  link_radius=0.05):
    """An ``rv_arm`` from a robot URDF (sawyer_sim.py:101-117 loads ARM_URDF the same way): the seven
    revolute joints of the chain base -> ``tip_link`` (origins, axes, limits, velocity and effort limits)
    and one collider box per limb link -- the box of its ``<collision>`` geometry in the joint frame (links
    behind fixed joints included); a link without collision geometry gets a box of ``link_radius`` around
    the segment to the next joint.  The gripper (base box, two prismatic fingers and their pads) keeps the
    geometry of ``scenes.make_arm``: RoboVat drives it through two joints off the hand link that are not
    on the chain.  ``max_accel``: joint acceleration limits (URDF has none), default scenes.SAWYER_MAX_ACCEL."""
    from robovat_amd import scenes
    ch = arm_chain_from_urdf(path, tip_link, base_link)
    if len(ch['names']) != abi.RV_NLIMB or any(k == 'prismatic' for k in ch['kind']):
        raise ValueError('the limb must be a chain of %d revolute joints, got %s' % (abi.RV_NLIMB, ch['kind']))
    arm = scenes.make_arm(base_pos=base_pos, base_rpy=base_rpy)           # gripper geometry, finger joints
    acc = list(max_accel) if max_accel is not None else list(scenes.SAWYER_MAX_ACCEL)
    for i in range(abi.RV_NLIMB):
        abi.assign(arm.jpos[i], ch['jpos'][i]); abi.assign(arm.jquat[i], ch['jquat'][i])
        arm.q_lo[i], arm.q_hi[i], arm.v_max[i], arm.a_max[i] = ch['q_lo'][i], ch['q_hi'][i], ch['v_max'][i], acc[i]
        arm.inv_tau_max[i] = 1.0 / ch['effort'][i] if ch['effort'][i] > 0 else 0.0
        pts = np.asarray(ch['colliders'][i])
        nxt = np.asarray(ch['jpos'][i + 1] if i + 1 < abi.RV_NLIMB else ch['tip']['pos'])
        if len(pts):
            lo, hi = pts.min(axis=0), pts.max(axis=0)
        else:
            lo, hi = np.minimum(0.0, nxt) - link_radius, np.maximum(0.0, nxt) + link_radius
        arm.col_frame[i] = i
        abi.assign(arm.col_center[i], (0.5 * (lo + hi)).tolist()); abi.assign(arm.col_half[i], (0.5 * (hi - lo)).tolist())
    abi.assign(arm.jpos[abi.RV_NLIMB], ch['tip']['pos']); abi.assign(arm.jquat[abi.RV_NLIMB], ch['tip']['quat'])
    # inertial parameters (PHYSICS.LIMB_DYNAMICS): per limb link the composite of the <inertial> elements riding on
    # its joint -- total mass, common centre of mass, and the DIAGONAL of the inertia tensor about it in the joint
    # frame (rv_arm keeps principal moments along the frame axes).  Links behind the last joint (hand, gripper base)
    # stay with the hand entry of scenes.make_arm unless the URDF has inertials for them.
    for i in range(abi.RV_NLIMB):
        parts = ch['inertials'][i]
        if not parts:
            continue
        if i == abi.RV_NLIMB - 1 and len(parts) > 1:       # wrist link + what is bolted to it: the link itself, and the hand entry
            groups = [(i, parts[:1]), (abi.RV_NLIMB, parts[1:])]
        else:
            groups = [(i, parts)]
        for slot, ps in groups:
            m = sum(p[0] for p in ps)
            com = sum(p[0] * np.asarray(p[1]) for p in ps) / m
            I = np.zeros((3, 3))
            for pm, pc, pI in ps:
                d = np.asarray(pc) - com
                I += np.asarray(pI) + pm * (d @ d * np.eye(3) - np.outer(d, d))
            if slot == abi.RV_NLIMB:                        # the hand frame: rotate / shift from the wrist joint frame
                Rt = _quat_matrix(ch['tip']['quat'])
                com = Rt.T @ (com - np.asarray(ch['tip']['pos'])); I = Rt.T @ I @ Rt
            arm.link_mass[slot] = m
            abi.assign(arm.link_com[slot], com.tolist()); abi.assign(arm.link_inertia[slot], np.diag(I).tolist())
    return arm


def fk_chain(chain, q):
    """World pose (position, 3x3 rotation) of the tip for joint values ``q`` (numpy check of the ingest;
    the same composition as the kernel's ``fk_chain``: parent . (jpos, jquat) . Rz(q) or Tz(q))."""
    p, R = np.zeros(3), np.eye(3)
    for i, qi in enumerate(q):
        p = p + R @ np.asarray(chain['jpos'][i])
        R = R @ _quat_matrix(chain['jquat'][i])
        if chain['kind'][i] == 'prismatic':
            p = p + R @ np.array([0.0, 0.0, qi])
        else:
            c, s = np.cos(qi), np.sin(qi)
            R = R @ np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    p = p + R @ np.asarray(chain['tip']['pos'])
    R = R @ _quat_matrix(chain['tip']['quat'])
    return p, R
