"""Procedural triangle meshes, for the large-mesh regime.

The reference's scenes/miata.json names an STL that its repository does not
ship (.MISSING_LARGE_BLOBS); these generators stand in for it: closed surfaces
with outward-facing triangles (the reference culls back faces,
Hittable.scala:103-110), any number of facets, written as arrays or as the
ASCII STL dialect BVH.fromSTL reads (BVH.scala:35-55).
"""
from __future__ import annotations

import numpy as np


def bumpy_ellipsoid(n_lat: int, n_lon: int, center=(-0.3, 2.05, -0.6), radii=(0.9, 2.0, 1.7),
                    bump: float = 0.08, waves=(9, 14)) -> np.ndarray:
    """[2 * n_lat * n_lon - 2 * n_lon, 3, 3] outward-facing triangles of an
    ellipsoid whose radius is modulated by 1 + bump * sin(waves[0] * theta) * sin(waves[1] * phi)."""
    theta = np.linspace(0.0, np.pi, n_lat + 1)            # pole to pole
    phi = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)        # around; column n_lon == column 0
    th, ph = np.meshgrid(theta, phi, indexing="ij")
    r = 1.0 + bump * np.sin(waves[0] * th) * np.sin(waves[1] * ph)
    p = np.stack([r * np.sin(th) * np.cos(ph), r * np.cos(th), r * np.sin(th) * np.sin(ph)], -1)
    p[:, -1] = p[:, 0]                                    # close the seam exactly
    p[0, :] = p[0, 0]                                     # one point per pole
    p[-1, :] = p[-1, 0]
    p = p * np.asarray(radii) + np.asarray(center)
    a, b = p[:-1, :-1], p[:-1, 1:]                        # upper edge of each quad
    c, d = p[1:, :-1], p[1:, 1:]                          # lower edge
    # with y up, theta growing downwards and phi turning x -> z, (a, b, c) and
    # (b, d, c) wind counter-clockwise seen from outside
    t1 = np.stack([a, b, c], -2)[1:]                      # degenerate at the top pole (a == b)
    t2 = np.stack([b, d, c], -2)[:-1]                     # degenerate at the bottom pole (c == d)
    tris = np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)])
    return np.ascontiguousarray(tris, dtype=np.float64)


def write_ascii_stl(path: str, tris: np.ndarray, name: str = "hiptrace meshgen") -> None:
    """ASCII STL with fixed-point numbers ("-?\\d+.\\d+": the reference's pattern has
    no exponent form) and the line structure BVH.fromSTL expects."""
    tris = np.asarray(tris, dtype=np.float64)
    n = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = n / np.where(ln > 0, ln, 1.0)
    with open(path, "w") as f:
        f.write(f"solid {name}\n")
        for k in range(len(tris)):
            f.write("facet normal %.6f %.6f %.6f\nouter loop\n" % tuple(n[k]))
            for c in range(3):
                f.write("vertex %.6f %.6f %.6f\n" % tuple(tris[k, c]))
            f.write("endloop\nendfacet\n")
        f.write(f"endsolid {name}\n")


def rounded(tris: np.ndarray, decimals: int = 6) -> np.ndarray:
    """The coordinates a reader of write_ascii_stl's output gets back."""
    return np.round(np.asarray(tris, dtype=np.float64), decimals)
