"""CPU restatement (numpy / pure Python) of the HOST-side algorithms of the depth-map stage — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product is the C++ code under alicevision_amd/host/.  It is written from the reference
sources, independently of the C++ host (different language, different data structures), and the tests compare the two on the
same scenes (tests/test_host_cpu.py).  PINNED to the reference's own code since round 2: tests/test_host_ref.py holds every function
below against oracle/_ref/libavdm_host_ref.so — depthMap/SgmDepthList.cpp and mvsUtils/TileParams.cpp compiled whole and unchanged,
the T-camera selection, the tile-merge weights and the projection / pixel-size / epipolar helpers compiled from the reference's text
(oracle/ref/Makefile) — which found and fixed two deviations (Pixel(Point2d) rounds half up; equal scores keep the order of the C
library's qsort).  Unpinned remainders, restated on both sides: boost::accumulators' tail quantile and camera::angleBetweenRays
(Eigen code over the camera model).  Paths relative to /root/reference/src/aliceVision.

    tile_roi_list               mvsUtils/TileParams.cpp:15-61
    tile_weight_map             mvsUtils/mapIO.cpp:170-311 (weightTileBorder / addSingleTileMapWeighted)
    Cameras                     mvsUtils/MultiViewParams.cpp:283-319 (P -> K, R, C by RQ), mvsData/Matrix3x4.hpp:80-114
    nearest_cams_from_landmarks mvsUtils/MultiViewParams.cpp:519-575
    tile_nearest_cams           mvsUtils/MultiViewParams.cpp:577-667
    depth_list                  depthMap/SgmDepthList.cpp:48-192, 277-660
"""
import math

import numpy as np

FLT_EPSILON = float(np.finfo(np.float32).eps)
f32 = np.float32


def ceil_div(a, b):
    return (a + b - 1) // b


def downscale_range(b, e, d):
    return int(math.floor(b / d)), int(math.ceil(e / d))


# ---------------------------------------------------------------------------------------------------------------- tiles
def tile_roi_list(buffer_w, buffer_h, padding, image_w, image_h, max_downscale):
    """list of (x0, x1, y0, y1) in process pixels"""
    if buffer_h >= image_w and buffer_h >= image_h:  # hasOnlyOneTile tests bufferHeight against both (TileParams.hpp:35-38)
        return [(0, image_w, 0, image_h)]
    nx = ceil_div(image_w, buffer_w - 2 * padding)
    ny = ceil_div(image_h, buffer_h - 2 * padding)
    etw = ceil_div(ceil_div(image_w, max_downscale), nx) * max_downscale
    eth = ceil_div(ceil_div(image_h, max_downscale), ny) * max_downscale
    out = []
    for i in range(nx):
        for j in range(ny):
            out.append((i * etw, min((i + 1) * etw + padding, image_w), j * eth, min((j + 1) * eth + padding, image_h)))
    return out


def tile_weight_map(roi, image_w, image_h, padding, downscale):
    """weights addSingleTileMapWeighted multiplies a tile with (tile resolution = roi / downscale)"""
    x0, x1, y0, y1 = roi
    bx, ex = downscale_range(x0, x1, downscale)
    by, ey = downscale_range(y0, y1, downscale)
    tw, th = ex - bx, ey - by
    pad = padding // downscale
    first_col, last_col, first_row, last_row = x0 == 0, x1 == image_w, y0 == 0, y1 == image_h
    wmap = np.ones((th, tw), np.float32)

    def border(a, b, c, d, bw, bh, lux, luy):
        rdx, rdy = lux + bw, luy + bh
        margin = 2.0
        bwm, bhm = bw - 2.0 * margin, bh - 2.0 * margin
        for x in range(int(lux), min(int(rdx), tw)):
            for y in range(int(luy), min(int(rdy), th)):
                r_x = f32(min(max((rdx - margin - x) / bwm, 0.0), 1.0))
                r_y = f32(min(max((rdy - margin - y) / bhm, 0.0), 1.0))
                l_x = f32(min(max((x - (lux + margin)) / bwm, 0.0), 1.0))
                l_y = f32(min(max((y - (luy + margin)) / bhm, 0.0), 1.0))
                w = r_y * (r_x * f32(a) + l_x * f32(b)) + l_y * (r_x * f32(d) + l_x * f32(c))
                wmap[y, x] *= f32(w)

    if not first_col or not first_row:
        border(0, 1 if first_row else 0, 1, 1 if first_col else 0, pad, pad, 0, 0)
    if not first_col or not last_row:
        border(1 if first_col else 0, 1, 1 if last_row else 0, 0, pad, pad, 0, th - pad)
    if not last_col or not first_row:
        border(1 if first_row else 0, 0, 1 if last_col else 0, 1, pad, pad, tw - pad, 0)
    if not last_col or not last_row:
        border(1, 1 if last_col else 0, 0, 1 if last_row else 0, pad, pad, tw - pad, th - pad)
    if not first_row:
        border(0, 0, 1, 1, tw - 2 * pad, pad, pad, 0)
    if not last_row:
        border(1, 1, 0, 0, tw - 2 * pad, pad, pad, th - pad)
    if not first_col:
        border(0, 1, 1, 0, pad, th - 2 * pad, 0, pad)
    if not last_col:
        border(1, 0, 0, 1, pad, th - 2 * pad, tw - pad, pad)
    return wmap, (bx, ex, by, ey)


# -------------------------------------------------------------------------------------------------------------- cameras
def _rq(H):
    """Matrix3x3::RQ (mvsData/Matrix3x3.hpp:193-265): Gram-Schmidt on the rows bottom-up"""
    a1, a2, a3 = H[2], H[1], H[0]
    e1 = a1 / np.linalg.norm(a1)
    u2 = a2 - e1 * (e1 @ a2) / (e1 @ e1)
    e2 = u2 / np.linalg.norm(u2)
    u3 = a3 - e1 * (e1 @ a3) / (e1 @ e1) - e2 * (e2 @ a3) / (e2 @ e2)
    e3 = u3 / np.linalg.norm(u3)
    Q = np.stack([e3, e2, e1])
    R = np.array([[e3 @ a3, e2 @ a3, e1 @ a3], [0.0, e2 @ a2, e1 @ a2], [0.0, 0.0, e1 @ a1]])
    return R, Q


def decompose_p(P):
    K, R = _rq(P[:, :3])
    K = K / abs(K[2, 2])
    if K[0, 0] < 0:
        D = np.diag([-1.0, -1.0, 1.0])
        K, R = K @ D, D @ R
    if K[1, 1] < 0:
        D = np.diag([1.0, -1.0, -1.0])
        K, R = K @ D, D @ R
    C = np.linalg.inv(-P[:, :3]) @ P[:, 3]
    return K, R, C


class Cameras:
    """cameras at process resolution from K (3x3), R_i, C_i of the synthetic scene"""

    def __init__(self, K, Rs, Cs, width, height, process_downscale=1, min_angle=2.0, max_angle=70.0):
        self.n = len(Rs)
        self.width, self.height = width // process_downscale, height // process_downscale
        self.ds = process_downscale
        self.min_angle, self.max_angle = f32(min_angle), f32(max_angle)
        self.P, self.K, self.R, self.C, self.iR, self.iCam = [], [], [], [], [], []
        self.K_full, self.R_full, self.C_full = K, Rs, Cs
        for i in range(self.n):
            P = K @ np.concatenate([Rs[i], (-Rs[i] @ Cs[i])[:, None]], axis=1)
            P = P.copy()
            P[:2] /= float(process_downscale)  # "for i < 8: m[i] /= scale" on a row-major 3x4 (MultiViewParams.cpp:289-291)
            Ki, Ri, Ci = decompose_p(P)
            self.P.append(P), self.K.append(Ki), self.R.append(Ri), self.C.append(Ci)
            self.iR.append(np.linalg.inv(Ri))
            self.iCam.append(np.linalg.inv(Ri) @ np.linalg.inv(Ki))

    def project(self, X, P):
        x = P[:, :3] @ X + P[:, 3]
        if x[2] <= 0:
            return np.array([-1.0, -1.0])
        return x[:2] / x[2]

    def pixel_in_image(self, pix, margin=2):
        px, py = int(math.floor(pix[0] + 0.5)), int(math.floor(pix[1] + 0.5))  # Pixel(Point2d), mvsData/Pixel.hpp:30-34: rounds half up
        return margin <= px < self.width - margin and margin <= py < self.height - margin

    def cam_pixel_size(self, x0, cam, d):
        if d == 0.0:
            return 0.0
        pix = self.project(x0, self.P[cam]).copy()
        pix[0] += d
        v = self.iCam[cam] @ np.array([pix[0], pix[1], 1.0])
        v /= np.linalg.norm(v)
        return np.linalg.norm(np.cross(v, self.C[cam] - x0))


def _ray(K, R, uv):
    c = np.array([(uv[0] - K[0, 2]) / K[0, 0], (uv[1] - K[1, 2]) / K[1, 1], 1.0])
    c /= np.linalg.norm(c)
    w = R.T @ c
    return w / np.linalg.norm(w)


def angle_between_rays(K, R1, R2, x1, x2):
    r1, r2 = _ray(K, R1, x1), _ray(K, R2, x2)
    c = min(max((r1 @ r2) / (np.linalg.norm(r1) * np.linalg.norm(r2)), -1.0 + 1e-8), 1.0 - 1e-8)
    return math.degrees(math.acos(c))


def _qsort_sorted_id_desc(ids):
    """qsort(ids, qsortCompareSortedIdDesc) as glibc executes it: the comparator (mvsData/structures.cpp:37-47) returns -1 when
    a.value > b.value and +1 otherwise — never 0 — and glibc's qsort is a top-down merge sort (stdlib/msort.c: halves n / 2 and n - n / 2,
    `cmp(b1, b2) <= 0` takes the element of the first half), so on equal values the element of the SECOND half goes first.
    ids: list of (id, value)."""
    n = len(ids)
    if n <= 1:
        return list(ids)
    a, b = _qsort_sorted_id_desc(ids[:n // 2]), _qsort_sorted_id_desc(ids[n // 2:])
    out, i, j = [], 0, 0
    while i < len(a) and j < len(b):
        if a[i][1] > b[j][1]:
            out.append(a[i])
            i += 1
        else:
            out.append(b[j])
            j += 1
    return out + a[i:] + b[j:]


def nearest_cams_from_landmarks(cams, landmarks, rc, nb):
    """landmarks: list of (X, {view index: (u, v) in FULL-size pixels}).  Equal scores keep the order the reference's qsort call gives
    them (_qsort_sorted_id_desc; pinned against the reference's own function, tests/test_host_ref.py)."""
    score = np.zeros(cams.n, np.float32)
    for _, obs in landmarks:
        if rc not in obs:
            continue
        for tc, x in obs.items():
            if tc == rc:
                continue
            a = angle_between_rays(cams.K_full, cams.R_full[rc], cams.R_full[tc], obs[rc], x)
            if a < cams.min_angle or a > cams.max_angle:
                continue
            score[tc] += 1
    order = [i for i, _ in _qsort_sorted_id_desc([(i, score[i]) for i in range(cams.n)])]
    return [i for i in order[:min(cams.n, nb)] if score[i] > 20]


def tile_nearest_cams(cams, landmarks, rc, nb, tcams, roi):
    def plateau(a, b, c, d, x):
        if a < x <= b:
            return f32(x - a) / f32(b - a)
        if b < x <= c:
            return f32(1.0)
        if c < x <= d:
            return f32(1.0) - f32(x - c) / f32(d - c)
        return f32(0.0)

    x0, x1, y0, y1 = roi
    fx0, fx1 = int(math.floor(x0 * cams.ds)), int(math.ceil(x1 * cams.ds))
    fy0, fy1 = int(math.floor(y0 * cams.ds)), int(math.ceil(y1 * cams.ds))
    score = {tc: f32(0.0) for tc in tcams}
    for _, obs in landmarks:
        if rc not in obs:
            continue
        u, v = int(obs[rc][0]), int(obs[rc][1])
        if not (fx0 <= u < fx1 and fy0 <= v < fy1):
            continue
        for tc, x in obs.items():
            if tc == rc or tc not in score:
                continue
            a = angle_between_rays(cams.K_full, cams.R_full[rc], cams.R_full[tc], obs[rc], x)
            score[tc] = f32(score[tc] + plateau(1, 10, 50, 150, int(a)))
    ids = _qsort_sorted_id_desc([(tc, s) for tc, s in sorted(score.items()) if s > 0])
    return [tc for tc, _ in ids[:min(cams.n, nb)]]


# ---------------------------------------------------------------------------------------------------------- depth list
def index_of_nearest_sorted(v, value):
    import bisect
    it = bisect.bisect_left(v, value)
    if it == len(v):
        return -1
    if it != 0 and (value - v[it - 1]) < (v[it] - value):
        it -= 1
    return it


def _tail_quantile(values, left, probability, cache=1000):
    """boost::accumulators tail_quantile<left|right> with tail cache size `cache` (published behaviour)"""
    n = int(math.ceil(len(values) * (probability if left else 1.0 - probability)))
    tail = sorted(values, reverse=not left)[:cache]
    if 0 < n < len(tail):
        return f32(tail[n - 1])
    return f32(np.nan)


def _line_line_intersect(p1, p2, p3, p4):
    p13, p43, p21 = p1 - p3, p4 - p3, p2 - p1
    if np.all(np.abs(p43) < FLT_EPSILON) or np.all(np.abs(p21) < FLT_EPSILON):
        return None
    d1343, d4321, d1321, d4343, d2121 = p13 @ p43, p43 @ p21, p13 @ p21, p43 @ p43, p21 @ p21
    denom = d2121 * d4343 - d4321 * d4321
    if abs(denom) < FLT_EPSILON:  # a NaN denominator passes, as in the reference
        return None
    mua = (d1343 * d4321 - d1321 * d4343) / denom
    mub = (d1343 + d4321 * mua) / d4343
    return ((p1 + mua * p21) + (p3 + mub * p43)) / 2.0


def _triangulate(cams, refpix, tarpix, rc, tc):
    rv = cams.iCam[rc] @ np.array([refpix[0], refpix[1], 1.0])
    rv = rv / np.linalg.norm(rv)
    tv = cams.iCam[tc] @ np.array([tarpix[0], tarpix[1], 1.0])
    tv = tv / np.linalg.norm(tv)
    return _line_line_intersect(cams.C[rc], cams.C[rc] + rv, cams.C[tc], cams.C[tc] + tv)


def _line_image_intersection(cams, lp1, lp2):
    v = lp2 - lp1
    pf, pt = np.zeros(2), np.zeros(2)
    if np.linalg.norm(v) < FLT_EPSILON:
        return pf, pt
    v = v / np.linalg.norm(v)
    a, b = -v[1], v[0]
    c = -a * lp1[0] - b * lp1[1]
    rw, rh = float(cams.width), float(cams.height)
    hits = []
    with np.errstate(divide="ignore", invalid="ignore"):
        for x, y, horizontal in ((0.0, np.float64(-c) / b, False), (rw, np.float64(-c - a * rw) / b, False), (np.float64(-c) / a, 0.0, True),
                                 (np.float64(-c - b * rh) / a, rh, True)):
            if (horizontal and 0 <= x < rw) or (not horizontal and 0 <= y < rh):
                hits.append(np.array([x, y], dtype=np.float64))
    if len(hits) >= 1:
        pf = hits[0]
    if len(hits) >= 2:
        pt = hits[-1]  # every later hit overwrites pTo
    if len(hits) == 2 and np.linalg.norm(lp1 - pf) > np.linalg.norm(lp1 - pt):
        pf, pt = pt, pf
    return pf, pt


def _angle_deg(v1, v2):
    a = math.acos(float((v1 / np.linalg.norm(v1)) @ (v2 / np.linalg.norm(v2))))
    return 0.0 if math.isnan(a) else abs(a / (math.pi / 180.0))


def depth_list(cams, landmarks, rc, tcams, roi, sgm_scale=2, max_depths=1500, step_z=-1, seeds_range_inflate=0.2, use_sfm_seeds=True,
               depth_list_per_tile=False, percentile=0.999, prematching_max_depth_scale=1.5):
    """returns (depths float32 list, [(first, count)] per T camera) or ([], []) — SgmDepthList::computeListRc"""
    x0, x1, y0, y1 = roi
    fx0, fx1 = int(math.floor(x0 * cams.ds)), int(math.ceil(x1 * cams.ds))
    fy0, fy1 = int(math.floor(y0 * cams.ds)), int(math.ceil(y1 * cams.ds))
    plane_p = cams.C[rc]
    n = cams.iR[rc] @ np.array([0.0, 0.0, 1.0])
    plane_n = n / np.linalg.norm(n)

    def plane_dist(X):
        return abs(X @ plane_n - plane_p @ plane_n) / math.sqrt(plane_n @ plane_n)

    def oriented(X):
        return (X @ plane_n - plane_p @ plane_n) / math.sqrt(plane_n @ plane_n)

    def in_roi(uv):
        return (not depth_list_per_tile) or (fx0 <= int(uv[0]) < fx1 and fy0 <= int(uv[1]) < fy1)

    dists, mid = [], np.zeros(3)
    for X, obs in landmarks:
        if rc in obs and in_roi(obs[rc]):
            dists.append(f32(plane_dist(X)))
            mid = mid + X
    nb_obs = len(dists)
    if nb_obs < 2:
        return [], []
    min_obs = _tail_quantile(dists, True, 1.0 - percentile)
    max_obs = _tail_quantile(dists, False, percentile)
    mid_obs = f32(plane_dist(mid / float(f32(nb_obs))))

    def rc_tc_depths(tc, mid_depth):
        ref = np.array([cams.width * 0.5, cams.height * 0.5]) if not depth_list_per_tile else np.array([x0 + (x1 - x0) * 0.5, y0 + (y1 - y0) * 0.5])
        K, R, C = decompose_p(cams.P[rc])
        riP = np.linalg.inv(R) @ np.linalg.inv(K)
        refh = np.array([ref[0], ref[1], 1.0])
        tc_mid = cams.project((riP @ refh) * float(mid_depth) + C, cams.P[tc])
        zs = [plane_dist(X) for X, obs in landmarks if tc in obs and rc in obs and in_roi(obs[rc])]
        assert zs, "no common observations"
        p1 = cams.project((riP @ refh) * min(zs) + C, cams.P[tc])
        p2 = cams.project((riP @ refh) * max(zs) + C, cams.P[tc])
        pf, pt = _line_image_intersection(cams, p1, p2)
        nb = int(np.linalg.norm(pt - pf)) // sgm_scale
        with np.errstate(invalid="ignore", divide="ignore"):
            pix_vect = (pt - pf) / np.linalg.norm(pt - pf) * max(1.0, float(sgm_scale))
        direction = 1
        p = _triangulate(cams, ref, tc_mid, rc, tc)
        if p is None:
            return []
        d0 = f32(oriented(p))
        p = _triangulate(cams, ref, tc_mid + pix_vect, rc, tc)
        if p is None:
            return []
        if d0 > f32(oriented(p)):
            direction = -1
        ref_vect = cams.iCam[rc] @ refh
        out, prev = [], f32(-1.0)
        for i in range(nb):
            tp = (pf if direction > 0 else pt) + pix_vect * float(i) * float(direction)
            if not cams.pixel_in_image(tp):
                continue
            ang = f32(_angle_deg(ref_vect, cams.iCam[tc] @ np.array([tp[0], tp[1], 1.0])))
            if ang < cams.min_angle or ang > cams.max_angle:
                continue
            p = _triangulate(cams, ref, tp, rc, tc)
            if p is None:
                continue
            d = f32(oriented(p))
            if d > 0 and d > prev:
                out.append(d)
                prev = f32(d + f32(FLT_EPSILON))
        return out

    def pixel_size_depths(min_d, mid_d, max_d):
        dd = float(sgm_scale) * 6.0
        maxdepth, k = f32(mid_d), 0
        while maxdepth < max_d and k < 1024:
            maxdepth = f32(maxdepth + f32(cams.cam_pixel_size(plane_p + plane_n * float(maxdepth), rc, dd)))
            k += 1
        mindepth, j = f32(mid_d), 0
        while mindepth > min_d and j < 2048 - k:
            mindepth = f32(mindepth - f32(cams.cam_pixel_size(plane_p + plane_n * float(mindepth), rc, dd)))
            j += 1
        out, depth, pix, cnt = [], f32(mindepth), f32(1.0), 0
        while depth < maxdepth and pix > 0 and cnt < 2048:
            out.append(depth)
            pix = f32(cams.cam_pixel_size(plane_p + plane_n * float(depth), rc, dd))
            depth = f32(depth + pix)
            cnt += 1
        return out

    per_tc = []
    for tc in tcams:
        d = rc_tc_depths(tc, -1 if nb_obs < 10 else mid_obs)
        if len(d) < 10:
            d = pixel_size_depths(min_obs, mid_obs, f32(max_obs * f32(prematching_max_depth_scale)))
        per_tc.append(d)
    all_d = [v for d in per_tc for v in d]
    if not all_d:
        return [], []
    min_all, max_all = min(all_d), max(all_d)
    first, last = f32(min_all), f32(max_all)
    if use_sfm_seeds and landmarks and nb_obs > 10:
        margin = f32(seeds_range_inflate * float(f32(max_obs - min_obs)))
        first, last = max(f32(0.0), f32(min_obs - margin)), f32(max_obs + margin)
        if not (max_all < first or min_all > last):
            first, last = max(f32(min_all), first), min(f32(max_all), last)

    def rc_depth_list(scale):
        out, depth = [], f32(first)
        while depth < last:
            out.append(depth)
            step = f32(last - first)
            for d in per_tc:
                i = index_of_nearest_sorted(d, depth)
                if i < 0 or i >= len(d) - 1:
                    continue
                step = min(step, f32(abs(f32(d[i] - d[i + 1]))))
            depth = f32(depth + f32(step * f32(scale)))
        return out

    depths = rc_depth_list(float(step_z) if step_z > 0 else 1.0)
    if max_depths > 0 and len(depths) > max_depths:
        depths = rc_depth_list(f32(f32(len(depths)) / f32(max_depths)))[:max_depths]
    limits = []
    for d in per_tc:
        i1, i2 = index_of_nearest_sorted(depths, d[0]), index_of_nearest_sorted(depths, d[-1])
        i1 = 0 if i1 == -1 else i1
        i2 = len(depths) - 1 if i2 == -1 else i2
        limits.append((i1, i2 - i1 + 1))
    return depths, limits
