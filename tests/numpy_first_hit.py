"""An independent numpy restatement of the reference's first-hit query, written from the
Scala sources alone (Camera.scala:5-23, Hittable.scala:4-8, 34-49, 55-92, 97-113) and
sharing no code with oracle/oracle.c.  Closest hit by a plain scan over all primitives:
what BVH.hit (BVH.scala:7-14) returns wherever two hits do not tie.  Test infrastructure
only; it exists so that a misreading of the Scala would have to be made twice, in two
languages and two program structures, to go unnoticed."""
import numpy as np

SPHERE, XYRECT, XZRECT, YZRECT, TRIANGLE = range(5)


def unit(v):
    return v / np.sqrt((v * v).sum())


def camera_rays(cam, W, H):
    """Camera.scala:5-23 with the lens closed and the pixel-centre offsets of the
    first-hit convention: u = (i + 0.5) / (W - 1), v = (j + 0.5) / (H - 1) (main.scala:88-89)."""
    origin, target, vup = (np.asarray(x, dtype=np.float64) for x in (cam.origin, cam.target, cam.vup))
    viewport_height = 2.0 * np.tan(np.radians(cam.vfov) / 2)
    viewport_width = cam.aspect_ratio * viewport_height
    w = unit(origin - target)
    u = unit(np.cross(vup, w))
    v = np.cross(w, u)
    focus = np.sqrt(((target - origin) ** 2).sum()) if cam.focus_distance < 0 else cam.focus_distance
    horizontal = focus * viewport_width * u
    vertical = focus * viewport_height * v
    lower_left_corner = origin - horizontal / 2 - vertical / 2 - focus * w
    s = (np.arange(W) + 0.5) / (W - 1)
    t = (np.arange(H) + 0.5) / (H - 1)
    direction = (lower_left_corner[None, None, :] + s[None, :, None] * horizontal[None, None, :]
                 + t[:, None, None] * vertical[None, None, :] - origin[None, None, :])
    return origin, direction  # direction [H, W, 3], not normalised


def camera_rays_fp32_arithmetic(cam, W, H):
    """The same rays as a device that holds the camera basis in fp32 and does the arithmetic in
    fp32 would make them: corner = float32(lower_left_corner - origin) (subtracted in fp64 first),
    s = float32(i + 0.5) / float32(W - 1), direction = fma(t, vertical, fma(s, horizontal, corner)).
    Returned as float64 arrays holding fp32 values.  (A fused multiply-add of floats is the
    float64 product, exact, plus the addend, rounded once to float64 and once more to float32.)"""
    origin, target, vup = (np.asarray(x, dtype=np.float64) for x in (cam.origin, cam.target, cam.vup))
    viewport_height = 2.0 * np.tan(np.radians(cam.vfov) / 2)
    viewport_width = cam.aspect_ratio * viewport_height
    w = unit(origin - target)
    u = unit(np.cross(vup, w))
    v = np.cross(w, u)
    focus = np.sqrt(((target - origin) ** 2).sum()) if cam.focus_distance < 0 else cam.focus_distance
    horizontal = focus * viewport_width * u
    vertical = focus * viewport_height * v
    lower_left_corner = origin - horizontal / 2 - vertical / 2 - focus * w
    f32 = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)
    corner, hor, ver, org = f32(lower_left_corner - origin), f32(horizontal), f32(vertical), f32(origin)
    s = (np.arange(W, dtype=np.float32) + np.float32(0.5)) / np.float32(W - 1)
    t = (np.arange(H, dtype=np.float32) + np.float32(0.5)) / np.float32(H - 1)
    s, t = s.astype(np.float64), t.astype(np.float64)
    inner = f32(s[None, :, None] * hor[None, None, :] + corner[None, None, :])
    direction = f32(t[:, None, None] * ver[None, None, :] + inner)
    return org, np.broadcast_to(direction, (H, W, 3)).copy()


def first_hit(cam, prim_type, prim_data, W, H, t_min=0.001, rays=None):
    """Returns t [H, W] (inf on a miss), prim id [H, W] (-1), Hit.normal [H, W, 3] and a
    mask of pixels whose two closest hits are too close to call.  rays: (origin [3],
    direction [H, W, 3]) to trace instead of the camera's own."""
    o, d = camera_rays(cam, W, H) if rays is None else rays
    best_t = np.full((H, W), np.inf)
    second_t = np.full((H, W), np.inf)
    best_id = np.full((H, W), -1, dtype=np.int32)
    best_n = np.zeros((H, W, 3))

    def take(k, t, ok, n):
        nonlocal best_t, second_t, best_id, best_n
        t = np.where(ok, t, np.inf)
        closer = t < best_t
        second_t = np.where(closer, best_t, np.minimum(second_t, t))
        best_t = np.where(closer, t, best_t)
        best_id = np.where(closer, k, best_id)
        best_n = np.where(closer[..., None], n, best_n)

    with np.errstate(divide="ignore", invalid="ignore"):
        for k, (kind, p) in enumerate(zip(prim_type, prim_data)):
            if kind == SPHERE:  # Hittable.scala:34-49
                center, radius = p[0:3], p[3]
                oc = o - center
                a = (d * d).sum(-1)
                half_b = (d * oc).sum(-1)
                c = (oc * oc).sum() - radius * radius
                disc = half_b * half_b - a * c
                root = np.sqrt(np.where(disc > 0, disc, 0.0))
                t1 = (-half_b - root) / a
                t2 = (-half_b + root) / a
                use1 = (t1 > t_min)  # t_max is +inf for a camera ray
                t = np.where(use1, t1, t2)
                ok = (disc > 0) & (t > t_min)
                n = ((o + t[..., None] * d) - center) / radius
            elif kind in (XYRECT, XZRECT, YZRECT):  # Hittable.scala:55-92
                ia, ib, ik = {XYRECT: (0, 1, 2), XZRECT: (0, 2, 1), YZRECT: (1, 2, 0)}[kind]
                a0, a1, b0, b1, kk = p[0:5]
                t = (kk - o[ik]) / d[..., ik]
                aa = o[ia] + t * d[..., ia]
                bb = o[ib] + t * d[..., ib]
                ok = ~((t < t_min) | (aa < a0) | (aa > a1) | (bb < b0) | (bb > b1)) & ~np.isnan(t)
                n = np.broadcast_to(np.array([0.0, 0.0, 1.0]), d.shape)  # the same for all three kinds
            else:  # Triangle, Hittable.scala:97-113
                A, B, C = p[0:3], p[3:6], p[6:9]
                edgeAB, edgeAC = B - A, C - A
                normal = np.cross(edgeAB, edgeAC)
                ao = o - A
                dao = np.cross(ao, d)
                det = -(d * normal).sum(-1)
                t = (ao * normal).sum() / det
                u = (dao * edgeAC).sum(-1) / det
                v = -(dao * edgeAB).sum(-1) / det
                wgt = 1 - u - v
                ok = (det > 1e-6) & (t >= t_min) & (u >= 0) & (v >= 0) & (wgt >= 0)
                n = np.broadcast_to(normal, d.shape)
            # Hit.normal: the outward normal turned to face the ray (Hittable.scala:6-7)
            front = (d * n).sum(-1) < 0
            take(k, t, ok, np.where(front[..., None], n, -n))
    with np.errstate(invalid="ignore"):
        tie = np.isfinite(best_t) & (second_t - best_t <= 1e-9 * np.maximum(1.0, np.abs(best_t)))
    return best_t, best_id, best_n, tie
