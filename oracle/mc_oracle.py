"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the iso-surface step that follows the hot path: xgutils/geoutil.py:175-233 `array2mesh`
(dim 3) = `mcubes.marching_cubes(grid, thresh)` + `verts/(Q-1)*(bbmax-bbmin)+bbmin`.

PARITY UNPINNED: PyMCubes 0.1.x (environment.yml:51) is a third-party C++ extension that is not under
/root/reference and not installed here, and the reference has no test or golden mesh for it.  What is restated is the
published algorithm it implements (Lorensen-Cline marching cubes: one vertex per sign-changing grid edge, linearly
interpolated; per-cell triangulation from the 8-corner inside/outside pattern).  Vertex/triangle ORDER and the
triangulation of ambiguous faces are implementation choices of PyMCubes that cannot be checked here; this oracle fixes
them as documented below, and tests assert the order-independent properties (closedness, Euler characteristic, enclosed
volume, vertices on the iso level) in addition to GPU == oracle.

Written independently of shapeformer_amd/mc_tables.py (no table: every cell is traced face by face), so that
tests/test_mcubes*.py can check the generated table against it for all 256 patterns.

Conventions (shared with csrc/mcubes.hip): inside = value > iso; ambiguous faces isolate inside corners; vertex order =
(grid point row-major, axis); triangle order = (cell row-major, loops by smallest cube-edge id, each fan-triangulated
from the first vertex - counted from the loop's smallest edge in surface direction - that puts no fan diagonal inside a
cube face, which would make that mesh edge non-manifold); normals point from inside to outside.
"""
import numpy as np

# the six faces as (axis, side); corner = (x, y, z) offsets.  Hand-written counter-clockwise cycles seen from outside.
_FACE_CYCLES = [
    [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0)],   # x = 0, outward -x
    [(1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)],   # x = 1, outward +x
    [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1)],   # y = 0, outward -y
    [(0, 1, 0), (0, 1, 1), (1, 1, 1), (1, 1, 0)],   # y = 1, outward +y
    [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0)],   # z = 0, outward -z
    [(0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)],   # z = 1, outward +z
]


def _check_ccw():
    for cyc in _FACE_CYCLES:
        p = np.array(cyc, float)
        n = np.cross(p[1] - p[0], p[2] - p[1])
        out = p.mean(0) - 0.5
        assert np.dot(n, out) > 0, cyc


_check_ccw()


def _edge_key(c0, c1):
    """cube edge between two adjacent corners -> (low corner offsets, axis) and the shared cube-edge id 4a+u+2v."""
    lo = tuple(min(a, b) for a, b in zip(c0, c1))
    axis = [i for i in range(3) if c0[i] != c1[i]][0]
    others = [i for i in range(3) if i != axis]
    return lo, axis, 4 * axis + lo[others[0]] + 2 * lo[others[1]]


def cell_loops(inside):
    """inside: dict corner offsets -> bool.  Directed loops of (low corner, axis, edge id)."""
    nxt, info = {}, {}
    for cyc in _FACE_CYCLES:
        flags = [inside[c] for c in cyc]
        for k in range(4):
            if flags[k] and not flags[(k + 1) % 4]:          # the walk leaves the inside set on edge k
                j = k
                while flags[j]:                                 # back over the inside run to the edge that entered it
                    j = (j - 1) % 4
                ef = _edge_key(cyc[k], cyc[(k + 1) % 4])
                et = _edge_key(cyc[j], cyc[(j + 1) % 4])
                nxt[ef[2]] = et[2]
                info[ef[2]], info[et[2]] = ef, et
    loops, seen = [], set()
    for s in sorted(nxt):
        if s in seen:
            continue
        loop, e = [], s
        while e not in seen:
            seen.add(e)
            loop.append(info[e])
            e = nxt[e]
        loops.append(loop)
    return loops


def _faces_of(lo, axis):
    """cube faces (axis, side) containing the cube edge that starts at corner `lo` and runs along `axis`."""
    return {(d, lo[d]) for d in range(3) if d != axis}


def _fan(ids, loop):
    """Fan triangulation from the first loop vertex none of whose diagonals lies inside a cube face (see module doc)."""
    n = len(ids)
    pick, pick_bad = 0, None
    for r in range(n):
        bad = 0
        for i in range(2, n - 1):
            (l0, a0, _), (l1, a1, _) = loop[r], loop[(r + i) % n]
            bad += bool(_faces_of(l0, a0) & _faces_of(l1, a1))
        if pick_bad is None or bad < pick_bad:
            pick, pick_bad = r, bad
        if bad == 0:
            break
    ids = ids[pick:] + ids[:pick]
    return [(ids[0], ids[k], ids[k + 1]) for k in range(1, n - 1)]


def _flip_needed():
    inside = {(x, y, z): (x, y, z) == (0, 0, 0) for x in (0, 1) for y in (0, 1) for z in (0, 1)}
    loop = cell_loops(inside)[0]
    p = []
    for lo, axis, _ in loop:
        q = np.array(lo, float)
        q[axis] += 0.5
        p.append(q)
    return np.dot(np.cross(p[1] - p[0], p[2] - p[0]), np.ones(3)) < 0


_FLIP = _flip_needed()


def marching_cubes(grid, iso=0.5, bbox=((-1, -1, -1), (1, 1, 1))):
    """grid (Q,Q,Q) -> verts (V,3) float32, faces (T,3) int32."""
    g = np.asarray(grid, np.float32)
    Q = g.shape[0]
    ins = g > np.float32(iso)
    vid, verts = {}, []
    lo_b, hi_b = np.asarray(bbox[0], np.float32), np.asarray(bbox[1], np.float32)
    for i0 in range(Q):
        for i1 in range(Q):
            for i2 in range(Q):
                for a in range(3):
                    j = [i0, i1, i2]
                    j[a] += 1
                    if j[a] >= Q or ins[i0, i1, i2] == ins[tuple(j)]:
                        continue
                    f0, f1 = g[i0, i1, i2], g[tuple(j)]
                    t = (np.float32(iso) - f0) / (f1 - f0)
                    pos = np.array([i0, i1, i2], np.float32)
                    pos[a] += t
                    vid[(i0, i1, i2, a)] = len(verts)
                    verts.append(pos / np.float32(Q - 1) * (hi_b - lo_b) + lo_b)
    faces = []
    for i0 in range(Q - 1):
        for i1 in range(Q - 1):
            for i2 in range(Q - 1):
                inside = {(x, y, z): bool(ins[i0 + x, i1 + y, i2 + z]) for x in (0, 1) for y in (0, 1) for z in (0, 1)}
                if all(inside.values()) or not any(inside.values()):
                    continue
                for loop in cell_loops(inside):
                    if _FLIP:
                        loop = loop[::-1]
                    ids = [vid[(i0 + lo[0], i1 + lo[1], i2 + lo[2], axis)] for lo, axis, _ in loop]
                    faces += _fan(ids, loop)
    return (np.array(verts, np.float32).reshape(-1, 3), np.array(faces, np.int32).reshape(-1, 3))


def pattern_triangles(ci):
    """Triangles (as cube-edge ids) of pattern ci (bit c set = corner (c&1, c>>1&1, c>>2&1) inside) - for the table check."""
    inside = {(x, y, z): bool((ci >> (x + 2 * y + 4 * z)) & 1) for x in (0, 1) for y in (0, 1) for z in (0, 1)}
    out = []
    for loop in cell_loops(inside):
        if _FLIP:
            loop = loop[::-1]
        out += _fan([e for _, _, e in loop], loop)
    return out


# ---- order-independent mesh properties used by the tests ---------------------------------------------------------------
def edge_use(faces):
    """directed-edge multiset check: closed, consistently oriented 2-manifold <=> every directed edge has its reverse."""
    f = np.asarray(faces, np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    key = e[:, 0] * (f.max() + 1) + e[:, 1]
    rkey = e[:, 1] * (f.max() + 1) + e[:, 0]
    return np.array_equal(np.sort(key), np.sort(rkey)) and len(np.unique(key)) == len(key)


def euler_characteristic(verts, faces):
    f = np.asarray(faces, np.int64)
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0), 1)
    E = len(np.unique(e[:, 0] * (f.max() + 1) + e[:, 1]))
    return len(verts) - E + len(f)


def signed_volume(verts, faces):
    v = np.asarray(verts, np.float64)
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    return float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)
