"""An independent numpy restatement of WHOLE camera paths of the reference, written from the
Scala sources alone and sharing no code with oracle/oracle.c:

  Camera.get_ray / ray_color      shared/src/main/scala/Camera.scala:19-36
  Hit                             shared/src/main/scala/Hittable.scala:4-8
  Sphere / XY,XZ,YZRect / Triangle  Hittable.scala:25-52, 54-94, 96-115
  Diffuse / Glossy / Glass / Emission  shared/src/main/scala/Material.scala:9-55
  SolidColor / Checker / Perlin / Image  shared/src/main/scala/Texture.scala:12-107
  the sample loop's u, v          jvm/src/main/scala/main.scala:88-89

The closest hit is a plain scan over all primitives (what BVH.hit, BVH.scala:7-14, returns
wherever two hits do not tie).  The reference draws its random numbers from an unseeded global
stream; here every random decision takes its uniforms from one Philox4x32-10 block per event,
counter (pixel, sample, event, 0), key = seed, 24-bit uniforms, with the uniform ball and disk
sampled directly from them (direction from u0, u1; ball radius cbrt(u2); disk radius sqrt(u2),
angle u3; Glass's reflectance test against u0): the same distributions as Vec3.scala:17-34 and
the convention the device kernels and the oracle's "matched" sampler use, so that the three
follow the same path.  Test infrastructure only: it exists so that a misreading of the Scala
would have to be made twice, in two languages and two program structures, to go unnoticed."""
import math

import numpy as np

SPHERE, XYRECT, XZRECT, YZRECT, TRIANGLE = range(5)
DIFFUSE, GLOSSY, GLASS, EMISSION = range(4)
SOLID, CHECKER, PERLIN, IMAGE = range(4)
M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (2011), Philox4x32 with ten rounds."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def uniforms(seed, pixel, sample, event):
    out = philox4x32_10((pixel, sample, event, 0), (seed & M32, (seed >> 32) & M32))
    return [(x >> 8) / 16777216.0 for x in out]


def unit(v):
    return v / math.sqrt(float(v @ v))


def unit_vector_from(u0, u1):  # Vec3.random_unit_vector (Vec3.scala:29-34) from two uniforms
    a = 2 * math.pi * u0
    z = 2.0 * u1 - 1.0
    r = math.sqrt(1 - z * z)
    return np.array([r * math.cos(a), r * math.sin(a), z])


class Paths:
    """The scene as the arrays that cross the C ABI (raytracing_amd.scene.FlatScene) and the
    camera of the scene file."""

    def __init__(self, cam, flat):
        self.flat = flat
        origin, target, vup = (np.asarray(x, dtype=np.float64) for x in (cam.origin, cam.target, cam.vup))
        viewport_height = 2.0 * math.tan(math.radians(cam.vfov) / 2)  # Camera.scala:5-17
        viewport_width = cam.aspect_ratio * viewport_height
        self.w = unit(origin - target)
        self.u = unit(np.cross(vup, self.w))
        self.v = np.cross(self.w, self.u)
        focus = math.sqrt(float((target - origin) @ (target - origin))) if cam.focus_distance < 0 else cam.focus_distance
        self.origin = origin
        self.horizontal = focus * viewport_width * self.u
        self.vertical = focus * viewport_height * self.v
        self.lower_left_corner = origin - self.horizontal / 2 - self.vertical / 2 - focus * self.w
        self.lens_radius = cam.aperture / 2
        self.background = np.asarray(cam.background, dtype=np.float64)
        pt, pd = np.asarray(flat.prim_type), np.asarray(flat.prim_data)
        self.sph = np.nonzero(pt == SPHERE)[0]
        self.tri = np.nonzero(pt == TRIANGLE)[0]
        self.rects = [(k, int(pt[k])) for k in np.nonzero((pt >= XYRECT) & (pt <= YZRECT))[0]]
        self.sc, self.sr = pd[self.sph, 0:3], pd[self.sph, 3]
        self.tA, self.tB, self.tC = pd[self.tri, 0:3], pd[self.tri, 3:6], pd[self.tri, 6:9]

    # ---- Hittable.hit of every primitive, closest accepted hit -----------------------------
    def closest_hit(self, o, d, t_min, t_max):
        """(t, primitive, outward_normal, uv) or None."""
        best = None
        pd = self.flat.prim_data
        with np.errstate(divide="ignore", invalid="ignore"):
            if len(self.sph):  # Hittable.scala:34-49
                oc = o[None, :] - self.sc
                a = float(d @ d)
                half_b = oc @ d
                c = (oc * oc).sum(-1) - self.sr * self.sr
                disc = half_b * half_b - a * c
                root = np.sqrt(np.where(disc > 0, disc, 0.0))
                t1 = (-half_b - root) / a
                t2 = (-half_b + root) / a
                use1 = (t1 < t_max) & (t1 > t_min)
                use2 = ~use1 & (t2 < t_max) & (t2 > t_min)
                t = np.where(use1, t1, np.where(use2, t2, np.inf))
                t = np.where(disc > 0, t, np.inf)
                k = int(np.argmin(t))
                if np.isfinite(t[k]):
                    n = ((o + t[k] * d) - self.sc[k]) / self.sr[k]
                    p1 = ((o + t1[k] * d) - self.sc[k]) / self.sr[k]  # get_uv of the t1 point in BOTH branches (:44, :46)
                    phi = math.atan2(p1[2], p1[0])
                    theta = math.asin(p1[1])
                    uv = (1 - (phi + math.pi) / (2 * math.pi), (theta + math.pi / 2) / math.pi)
                    best = (float(t[k]), int(self.sph[k]), n, uv)
            if len(self.tri):  # Hittable.scala:97-113
                eAB, eAC = self.tB - self.tA, self.tC - self.tA
                nv = np.cross(eAB, eAC)
                ao = o[None, :] - self.tA
                dao = np.cross(ao, d[None, :])
                det = -(nv @ d)
                t = (ao * nv).sum(-1) / det
                uu = (eAC * dao).sum(-1) / det
                vv = -(eAB * dao).sum(-1) / det
                ww = 1 - uu - vv
                lim = t_max if best is None else min(t_max, best[0])
                ok = (det > 1e-6) & (t >= t_min) & (t <= lim) & (uu >= 0) & (vv >= 0) & (ww >= 0)
                t = np.where(ok, t, np.inf)
                k = int(np.argmin(t))
                if np.isfinite(t[k]) and (best is None or t[k] < best[0]):
                    best = (float(t[k]), int(self.tri[k]), nv[k].copy(), (float(uu[k]), float(vv[k])))
        for k, kind in self.rects:  # Hittable.scala:55-92
            ia, ib, ik = {XYRECT: (0, 1, 2), XZRECT: (0, 2, 1), YZRECT: (1, 2, 0)}[kind]
            a0, a1, b0, b1, kk = pd[k, 0:5]
            if d[ik] == 0.0:
                continue
            t = (kk - o[ik]) / d[ik]
            lim = t_max if best is None else min(t_max, best[0])
            if t < t_min or t > lim:
                continue
            aa, bb = o[ia] + t * d[ia], o[ib] + t * d[ib]
            if aa < a0 or aa > a1 or bb < b0 or bb > b1:
                continue
            if best is None or t < best[0]:
                best = (float(t), int(k), np.array([0.0, 0.0, 1.0]), ((aa - a0) / (a1 - a0), (bb - b0) / (b1 - b0)))
        return best

    # ---- Texture.apply ------------------------------------------------------------------------
    def texture(self, tex, uv, p):
        f = self.flat
        kind = int(f.a_tex_type[tex])
        if kind == SOLID:
            return np.array(f.a_tex_data[tex, 0:3])
        if kind == CHECKER:  # Texture.scala:20-25
            size = f.a_tex_data[tex, 0]
            sines = math.sin(size * p[0]) * math.sin(size * p[1]) * math.sin(size * p[2])
            return self.texture(int(f.a_tex_child[tex, 0 if sines < 0 else 1]), uv, p)
        if kind == PERLIN:  # Texture.scala:90
            scale, table = f.a_tex_data[tex, 0], int(f.a_tex_child[tex, 0])
            g = 0.5 * (1 + math.sin(scale * p[2] + 10 * self.turb(table, p)))
            return np.array([g, g, g])
        # Image, Texture.scala:97-107; texels are fromRGB of 0x00RRGGBB words (Vec3.scala:36-40)
        im = int(f.a_tex_child[tex, 0])
        width, height, off = int(f.a_img_w[im]), int(f.a_img_h[im]), int(f.a_img_off[im])
        u = min(max(uv[0], 0.0), 1.0)
        v = 1.0 - min(max(uv[1], 0.0), 1.0)
        i = min(max(int(u * width), 0), width - 1)
        j = min(max(int(v * height), 0), height - 1)
        rgb = int(f.a_texels[off + j * width + i])
        return np.array([((rgb >> 16) & 0xFF) / 255.0, ((rgb >> 8) & 0xFF) / 255.0, (rgb & 0xFF) / 255.0])

    def noise(self, table, p):  # Texture.scala:44-76
        ranvec, perm = self.flat.a_ranvec[table], self.flat.a_perm[table]
        fl = [math.floor(x) for x in p]
        u, v, w = (p[0] - fl[0], p[1] - fl[1], p[2] - fl[2])
        i, j, k = (int(x) for x in fl)
        uu, vv, ww = u * u * (3 - 2 * u), v * v * (3 - 2 * v), w * w * (3 - 2 * w)
        total = 0.0
        for di in range(2):
            for dj in range(2):
                for dk in range(2):
                    c = ranvec[int(perm[0][(i + di) & 255]) ^ int(perm[1][(j + dj) & 255]) ^ int(perm[2][(k + dk) & 255])]
                    weight_v = np.array([u - di, v - dj, w - dk])
                    total += ((di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv))
                              * (dk * ww + (1 - dk) * (1 - ww)) * float(c @ weight_v))
        return total

    def turb(self, table, p, depth=7):  # Texture.scala:78-88
        accum, temp_p, weight = 0.0, np.array(p, dtype=np.float64), 1.0
        for _ in range(depth):
            accum += weight * self.noise(table, temp_p)
            temp_p = temp_p * 2
            weight *= 0.5
        return abs(accum)

    # ---- Material.scatter / emit --------------------------------------------------------------
    def scatter(self, mat, ray_o, ray_d, position, normal, front_face, uv, u4):
        """((origin, direction), attenuation) or None, and the emitted radiance."""
        f = self.flat
        kind, tex, param = int(f.a_mat_type[mat]), int(f.a_mat_texture[mat]), float(f.a_mat_param[mat])
        zero = np.zeros(3)
        if kind == EMISSION:  # Material.scala:52-55
            return None, param * self.texture(tex, uv, position)
        if kind == DIFFUSE:  # Material.scala:9-13
            return ((position, normal + unit_vector_from(u4[0], u4[1])), self.texture(tex, uv, position)), zero
        ud = unit(ray_d)
        if kind == GLOSSY:  # Material.scala:15-21
            in_sphere = (u4[2] ** (1.0 / 3.0)) * unit_vector_from(u4[0], u4[1])
            direction = (ud - 2 * float(ud @ normal) * normal) + param * in_sphere
            if float(direction @ normal) > 0:
                return ((position, direction), self.texture(tex, uv, position)), zero
            return None, zero
        # Glass, Material.scala:23-50
        etai_over_etat = 1.0 / param if front_face else param
        cos_theta = min(-float(ud @ normal), 1.0)
        sin_theta = math.sqrt(1.0 - cos_theta * cos_theta)
        r0 = (1 - etai_over_etat) / (1 + etai_over_etat)
        r1 = r0 * r0
        schlick = r1 + (1 - r1) * (1 - cos_theta) ** 5
        if etai_over_etat * sin_theta > 1 or schlick > u4[0]:
            out = ud - 2 * float(ud @ normal) * normal
        else:
            ct = -float(ud @ normal)
            perp = etai_over_etat * (ud + ct * normal)
            out = perp + (-math.sqrt(abs(1.0 - float(perp @ perp)))) * normal
        return ((position, out), self.texture(tex, uv, position)), zero

    # ---- Camera.ray_color, recursively as the reference writes it ---------------------------------
    def ray_color(self, o, d, depth, event, key):
        if depth <= 0:
            return np.zeros(3)
        hit = self.closest_hit(o, d, 0.001, math.inf)
        if hit is None:
            return self.background
        t, prim, outward, uv = hit
        position = o + t * d
        front_face = float(d @ outward) < 0
        normal = outward if front_face else -outward
        u4 = uniforms(key[0], key[1], key[2], event)
        scattered, emit = self.scatter(int(self.flat.prim_material[prim]), o, d, position, normal, front_face, uv, u4)
        if scattered is None:
            return emit
        (so, sd), attenuation = scattered
        return emit + attenuation * self.ray_color(so, sd, depth - 1, event + 1, key)

    def path_radiance(self, W, H, i, j, sample, seed=0, max_depth=50):
        """One sample of pixel (i, j), j = 0 the bottom row (main.scala:88-90)."""
        pixel = j * W + i
        u4 = uniforms(seed, pixel, sample, 0)
        s = (i + u4[0]) / (W - 1)
        t = (j + u4[1]) / (H - 1)
        r = math.sqrt(u4[2])
        phi = 2 * math.pi * u4[3]
        rd = self.lens_radius * np.array([r * math.cos(phi), r * math.sin(phi), 0.0])  # Camera.scala:19-23
        offset = self.u * rd[0] + self.v * rd[1]
        o = self.origin + offset
        d = self.lower_left_corner + s * self.horizontal + t * self.vertical - self.origin - offset
        return self.ray_color(o, d, max_depth, 1, (seed, pixel, sample))
