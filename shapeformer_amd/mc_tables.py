"""Marching-cubes case table for csrc/mcubes.hip, generated (not transcribed) from the cube topology.

The reference extracts its meshes with PyMCubes (`mcubes.marching_cubes`, third party, xgutils/geoutil.py:199),
i.e. classic Lorensen-Cline marching cubes.  The table here is built from first principles:

  corner c in 0..7 sits at offsets (c & 1, (c >> 1) & 1, (c >> 2) & 1) along grid axes (0, 1, 2);
  edge   e = 4 a + (u + 2 v) runs along axis a from the corner whose two OTHER offsets (in increasing axis order)
           are (u, v);
  a corner is "inside" when its sample is > iso.  On every cube face (seen from outside, corners counter-clockwise)
  the cut edges are joined by directed segments from the edge where the boundary walk LEAVES the inside set to the
  edge where it ENTERS it; a face with four cut edges isolates each inside corner (the same undirected choice is
  made by the neighbouring cell, so the surface is crack-free).  Every cut edge then has one outgoing and one
  incoming segment, the segments close into directed loops, and each loop is fan-triangulated from the first vertex
  (in loop order from its smallest edge id) whose fan has no diagonal inside a cube face.  The orientation is fixed so
  that triangle normals point from inside to outside.

`tables()` -> (ntri (256,) uint8, tri (256, MAXT*3) uint8 edge ids, 255 = unused).
`emit_inc(path)` writes the C initialiser that csrc/mcubes.hip includes.
"""
from __future__ import annotations

import numpy as np

MAXT = 5


def corner_offset(c):
    return (c & 1, (c >> 1) & 1, (c >> 2) & 1)


def edge_id(a, u, v):
    return 4 * a + u + 2 * v


def edge_corners(e):
    """(corner at the low end, corner at the high end) of edge e."""
    a, uv = divmod(e, 4)
    u, v = uv & 1, uv >> 1
    others = [x for x in range(3) if x != a]
    off = [0, 0, 0]
    off[others[0]], off[others[1]] = u, v
    lo = off[0] + 2 * off[1] + 4 * off[2]
    return lo, lo + (1 << a)


def _edge_between(c0, c1):
    d = c0 ^ c1
    a = {1: 0, 2: 1, 4: 2}[d]
    lo = min(c0, c1)
    off = corner_offset(lo)
    others = [x for x in range(3) if x != a]
    return edge_id(a, off[others[0]], off[others[1]])


def _faces():
    """Six faces as corner 4-cycles, counter-clockwise when seen from outside the cube."""
    faces = []
    for a in range(3):
        b, c = (a + 1) % 3, (a + 2) % 3          # (a, b, c) is a cyclic (right-handed) permutation of the axes
        for side in (0, 1):
            def corner(pb, pc):
                off = [0, 0, 0]
                off[a], off[b], off[c] = side, pb, pc
                return off[0] + 2 * off[1] + 4 * off[2]
            cyc = [corner(0, 0), corner(1, 0), corner(1, 1), corner(0, 1)]   # ccw seen from +a
            if side == 0:
                cyc = cyc[::-1]                                               # outward normal is -a
            faces.append(cyc)
    return faces


FACES = _faces()


def case_loops(cube_index):
    """Directed vertex loops (lists of edge ids) of one of the 256 inside/outside patterns."""
    inside = [(cube_index >> c) & 1 for c in range(8)]
    nxt = {}
    for cyc in FACES:
        leave, enter = [], []        # positions k of face edges (c_k -> c_{k+1}) that leave / enter the inside set
        for k in range(4):
            i0, i1 = inside[cyc[k]], inside[cyc[(k + 1) % 4]]
            if i0 and not i1:
                leave.append(k)
            elif i1 and not i0:
                enter.append(k)
        for kl in leave:
            # walk BACKWARDS from the leave edge over the inside corners to the enter edge that opened this inside run:
            # pairing them isolates each inside corner on a four-cut face
            k = kl
            while True:
                k = (k - 1) % 4
                if k in enter:
                    break
            e_from = _edge_between(cyc[kl], cyc[(kl + 1) % 4])
            e_to = _edge_between(cyc[k], cyc[(k + 1) % 4])
            assert e_from not in nxt
            nxt[e_from] = e_to
    loops, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, e = [], start
        while e not in seen:
            seen.add(e)
            loop.append(e)
            e = nxt[e]
        assert e == start
        loops.append(loop)
    return loops


def edge_faces(e):
    """The two cube faces (axis, side) that contain edge e."""
    a, uv = divmod(e, 4)
    others = [x for x in range(3) if x != a]
    return {(others[0], uv & 1), (others[1], uv >> 1)}


def fan_order(loop):
    """Rotation of `loop` whose fan has no diagonal lying inside a cube face (both cut edges on one face): such a
    diagonal would coincide with a diagonal of the neighbouring cell and make the edge non-manifold.  First rotation
    that works; loops of 3 or 4 vertices on distinct faces never need it."""
    n = len(loop)
    best, best_bad = loop, None
    for r in range(n):
        rot = loop[r:] + loop[:r]
        bad = sum(1 for i in range(2, n - 1) if edge_faces(rot[0]) & edge_faces(rot[i]))
        if best_bad is None or bad < best_bad:
            best, best_bad = rot, bad
        if bad == 0:
            break
    return best, best_bad


def _edge_mid(e):
    lo, hi = edge_corners(e)
    return (np.array(corner_offset(lo), float) + np.array(corner_offset(hi), float)) / 2


def _orientation_sign():
    """+1 if the loops as traced give normals pointing inside -> outside, else -1 (decided on the one-corner case)."""
    loop = case_loops(1)[0]                      # corner 0 inside: a single triangle around the origin corner
    p = [_edge_mid(e) for e in loop]
    n = np.cross(p[1] - p[0], p[2] - p[0])
    return 1 if np.dot(n, np.ones(3)) > 0 else -1    # outside is the (+,+,+) direction from corner 0


_SIGN = None


def tables():
    global _SIGN
    if _SIGN is None:
        _SIGN = _orientation_sign()
    ntri = np.zeros(256, np.uint8)
    tri = np.full((256, MAXT * 3), 255, np.uint8)
    for ci in range(256):
        t = []
        for loop in case_loops(ci):
            if _SIGN < 0:
                loop = loop[::-1]
            loop, _ = fan_order(loop)
            for i in range(1, len(loop) - 1):
                t += [loop[0], loop[i], loop[i + 1]]
        assert len(t) <= MAXT * 3, (ci, len(t))
        ntri[ci] = len(t) // 3
        tri[ci, :len(t)] = t
    return ntri, tri


def emit_inc(path):
    ntri, tri = tables()
    with open(path, "w") as f:
        f.write("// generated by shapeformer_amd/mc_tables.py (emit_inc) - do not edit\n")
        f.write(f"#define MC_MAXT {MAXT}\n")
        f.write("__device__ __constant__ unsigned char MC_NTRI[256] = {" + ",".join(str(int(x)) for x in ntri) + "};\n")
        f.write("__device__ __constant__ unsigned char MC_TRI[256][%d] = {\n" % (MAXT * 3))
        for ci in range(256):
            f.write("  {" + ",".join(str(int(x)) for x in tri[ci]) + "},\n")
        f.write("};\n")


if __name__ == "__main__":
    import os
    emit_inc(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "mc_table.h"))
    n, _ = tables()
    print("max triangles per cell:", int(n.max()), " total:", int(n.sum()))
