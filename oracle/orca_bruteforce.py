"""oracle/orca_bruteforce.py -- TEST INFRASTRUCTURE.  An INDEPENDENT float64 statement of what ORCA must compute,
written from the paper (van den Berg, Guy, Lin, Manocha: "Reciprocal n-body collision avoidance", sections 4-5), NOT from
oracle/orca_ref.h: it shares no code path, no branch structure and no incremental algorithm with the restatement of
RVO2 that the HIP kernel is bit-identical to.  It pins the SEMANTICS of that restatement (the upstream rvo2 source is
absent from /root/reference):

  * half-planes: the truncated velocity obstacle VO^tau_{A|B} is the union over t in (0, tau] of the discs
    D((p_B - p_A)/t, (r_A + r_B)/t): a cut-off disc D(p/tau, R/tau) plus the cone spanned by its tangent legs.  u is the
    vector from v_A - v_B to the CLOSEST POINT OF THE BOUNDARY of that set (found here by taking the minimum over the
    three boundary pieces), n the outward normal there; ORCA = { v : (v - (v_A + c u)) . n >= 0 } with c = 1/2
    (the fork's collab coefficient).  Already-colliding pairs use the cut-off disc of tau = time step only (RVO2).
  * feasible case: the new velocity is THE point of (intersection of half-planes) n (disc |v| <= v_max) closest to the
    preferred velocity -- found here by vertex / edge enumeration of the arrangement, no incremental LP.
  * infeasible case: RVO2's linearProgram3 returns a point of the disc that minimises the maximum penetration
    max_i (signed distance into the forbidden side of line i) -- the min-max VALUE is found here by enumerating the
    candidate optima of a convex piecewise-linear function on a disc.
"""
import itertools

import numpy as np


def _unit(v):
    return v / np.hypot(v[0], v[1])


def half_plane(pa, va, ra, pb, vb, rb, tau, dt, collab=0.5):
    """-> (point, normal, margin): permitted velocities v satisfy (v - point) . normal >= 0.  margin = gap between the
    best and the second-best boundary piece (small margin = geometrically ambiguous configuration)."""
    p = np.asarray(pb, float) - np.asarray(pa, float)
    v = np.asarray(va, float) - np.asarray(vb, float)
    R = ra + rb
    d = np.hypot(p[0], p[1])
    cands = []  # (distance to the boundary piece, closest point, outward normal)
    if d > R:
        c, rc = p / tau, R / tau
        w = v - c
        wl = np.hypot(w[0], w[1])
        # tangent legs: rays from the tangent points of the cut-off disc, directions = the tangents from the origin to
        # the disc D(p, R) (the same for every t)
        phi = np.arcsin(R / d)
        base = np.arctan2(p[1], p[0])
        leg_len0 = np.sqrt(d * d - R * R) / tau           # distance origin -> tangent point of the cut-off disc
        for sgn in (+1.0, -1.0):
            ang = base + sgn * phi
            t = np.array([np.cos(ang), np.sin(ang)])      # leg direction (away from the origin)
            s = max(float(v @ t), leg_len0)               # clamp to the ray's start
            q = s * t
            n_out = np.array([-t[1], t[0]]) * sgn         # points away from the cone's axis
            cands.append((np.hypot(*(v - q)), q, n_out))
        # cut-off arc: the part of the circle facing the origin, between the two tangent points
        if wl > 0:
            q = c + rc * w / wl
            # on the arc iff the angle between w and -p is below the tangent point's angle (cos = R / d)
            if (w @ (-p)) / (wl * d) >= R / d:
                cands.append((abs(wl - rc), q, w / wl))
    else:
        c, rc = p / dt, R / dt
        w = v - c
        wl = np.hypot(w[0], w[1])
        q = c + rc * w / wl
        cands.append((abs(wl - rc), q, w / wl))
    cands.sort(key=lambda x: x[0])
    dist, q, n_out = cands[0]
    margin = (cands[1][0] - dist) if len(cands) > 1 else np.inf
    u = q - v
    point = np.asarray(va, float) + collab * u
    return point, n_out, margin


def penetration(points, normals, v):
    """signed distance of v into the forbidden side of every line (> 0: violated)"""
    return -np.einsum("ij,ij->i", v[None, :] - points, normals)


def _circle_line(point, direction, radius):
    """intersections of the line point + t direction (|direction| = 1) with the circle |v| = radius"""
    b = float(point @ direction)
    disc = b * b - float(point @ point) + radius * radius
    if disc < 0:
        return []
    s = np.sqrt(disc)
    return [point + (-b - s) * direction, point + (-b + s) * direction]


def solve(points, normals, pref, vmax, eps=1e-9):
    """-> dict(feasible, v, dist, minmax): brute force over the arrangement of the lines and the speed circle."""
    points, normals = np.asarray(points, float).reshape(-1, 2), np.asarray(normals, float).reshape(-1, 2)
    n = len(points)
    dirs = np.stack([normals[:, 1], -normals[:, 0]], axis=1)  # line directions
    pref = np.asarray(pref, float)
    cand = []
    pl = np.hypot(pref[0], pref[1])
    cand.append(pref if pl <= vmax else pref * (vmax / pl))
    for i in range(n):
        t = float((pref - points[i]) @ dirs[i])
        cand.append(points[i] + t * dirs[i])              # projection of pref on line i
        cand.extend(_circle_line(points[i], dirs[i], vmax))
    for i, j in itertools.combinations(range(n), 2):
        A = np.array([normals[i], normals[j]])
        if abs(np.linalg.det(A)) > 1e-12:
            cand.append(np.linalg.solve(A, np.array([normals[i] @ points[i], normals[j] @ points[j]])))
    cand = np.array(cand)
    ok = (np.hypot(cand[:, 0], cand[:, 1]) <= vmax + eps)
    if n:
        pen = -(np.einsum("cj,ij->ci", cand, normals) - np.einsum("ij,ij->i", points, normals)[None, :])
        ok &= (pen <= eps).all(axis=1)
    out = {"feasible": bool(ok.any())}
    if ok.any():
        dist = np.hypot(*(cand[ok] - pref).T)
        k = int(np.argmin(dist))
        out["v"], out["dist"] = cand[ok][k], float(dist[k])
    out["minmax"] = minmax_penetration(points, normals, vmax) if n else -np.inf
    return out


def minmax_penetration(points, normals, vmax):
    """min over the disc |v| <= vmax of max_i penetration_i(v): the optimum of a convex piecewise-linear function on a
    disc is attained where (a) one function is minimal on the disc, (b) two functions are equal on the circle or at the
    point of their equality line closest to ... (covered by (c) and the circle), (c) three functions are equal."""
    n = len(points)
    off = np.einsum("ij,ij->i", points, normals)          # pen_i(v) = off_i - n_i . v
    cand = [np.zeros(2)]
    for i in range(n):
        cand.append(vmax * normals[i])                    # minimises pen_i on the disc
    for i, j in itertools.combinations(range(n), 2):
        dn = normals[i] - normals[j]                      # pen_i = pen_j  <=>  dn . v = off_i - off_j
        l = np.hypot(dn[0], dn[1])
        if l < 1e-12:
            continue
        nn = dn / l
        p0 = nn * ((off[i] - off[j]) / l)
        dd = np.array([nn[1], -nn[0]])
        cand.extend(_circle_line(p0, dd, vmax))
        cand.append(p0)
        # along the equality line both functions change linearly: the best point is at an end (circle) or where a third
        # function takes over (triples below)
    for i, j, k in itertools.combinations(range(n), 3):
        A = np.array([normals[i] - normals[j], normals[i] - normals[k]])
        if abs(np.linalg.det(A)) > 1e-12:
            cand.append(np.linalg.solve(A, np.array([off[i] - off[j], off[i] - off[k]])))
    cand = np.array(cand)
    cand = cand[np.hypot(cand[:, 0], cand[:, 1]) <= vmax + 1e-9]
    pen = off[None, :] - cand @ normals.T
    return float(pen.max(axis=1).min())
