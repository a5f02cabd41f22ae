"""Procedural stand-in for the mesh of the reference's `bust` scene (main.rs:1326-1371 loads models/Emperor.obj, or
models/David.obj with 766 484 triangles; neither file is part of the reference's repository): a closed, bumpy
head-and-shoulders surface of revolution written as a Wavefront .obj with the same number of triangles, so that
`load_model` and the BVH builder see a mesh of that size.  Plumbing for benchmarks and tests, not part of the path."""
import os
import tempfile

import numpy as np


def bust_obj(path, triangles=766_484):
    """Writes a mesh of about `triangles` triangles (exactly 2 * n_lon * n_lat) to `path`; returns the exact count."""
    n_lon = max(8, int(round((triangles / 2.0) ** 0.5)))
    n_lat = max(4, int(round(triangles / 2.0 / n_lon)))
    # profile: radius as a function of height, shoulders -> neck -> head (y in [-70, 70], about 120 across)
    t = np.linspace(0.0, 1.0, n_lat + 1)
    y = -70.0 + 140.0 * t
    shoulders = 58.0 * np.clip(1.0 - t / 0.38, 0.0, 1.0) ** 0.6
    neck = 14.0 + 0.0 * t
    head = 30.0 * np.sqrt(np.clip(1.0 - ((t - 0.72) / 0.28) ** 2, 0.0, 1.0))
    r = np.maximum(np.maximum(shoulders, head), np.where(t < 0.74, neck, 0.0))
    r[0] = max(r[0], 1.0)
    r[-1] = 0.02
    phi = np.linspace(0.0, 2.0 * np.pi, n_lon, endpoint=False)
    # bumps that break the symmetry: a face-like bulge, waves, fine ripples (all smooth, so triangles stay well shaped)
    bump = (1.0 + 0.10 * np.cos(phi)[None, :] * np.exp(-((t - 0.7) / 0.12) ** 2)[:, None]
            + 0.03 * np.sin(7.0 * phi)[None, :] * np.sin(11.0 * np.pi * t)[:, None]
            + 0.01 * np.sin(61.0 * phi)[None, :] * np.cos(83.0 * np.pi * t)[:, None])
    rr = r[:, None] * bump
    x = rr * np.cos(phi)[None, :]
    z = rr * np.sin(phi)[None, :]
    yy = np.repeat(y[:, None], n_lon, axis=1)
    verts = np.stack([x, yy, z], axis=2).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = i * n_lon + j + 1                       # .obj indices are 1-based
    b_ = i * n_lon + (j + 1) % n_lon + 1
    c = (i + 1) * n_lon + (j + 1) % n_lon + 1
    d = (i + 1) * n_lon + j + 1
    faces = np.concatenate([np.stack([a, b_, c], axis=2).reshape(-1, 3), np.stack([a, c, d], axis=2).reshape(-1, 3)])
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as fh:
        fh.write(f"# procedural bust, {len(faces)} triangles\n")
        np.savetxt(fh, verts, fmt="v %.9g %.9g %.9g")
        np.savetxt(fh, faces, fmt="f %d %d %d")
    os.replace(tmp, path)
    return len(faces)


def cached_bust(triangles=766_484):
    """Path of the generated mesh, written on first use under the system's temporary directory."""
    path = os.path.join(tempfile.gettempdir(), f"rt_amd_bust_{triangles}.obj")
    if not os.path.exists(path):
        bust_obj(path, triangles)
    return path
