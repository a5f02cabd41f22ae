#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/.

1. reference_cylinders_blockmeans.npz -- statistics of the one image in the reference tree that is a render of a
   benchmark scene at known settings: images/cornell_box_two_cylinders_medium_quality.png (cornell_quadric, 600x600,
   200 spp, depth 50; main.rs:851-852).  Only derived numbers are stored (6x6 block means of the 8-bit image, global
   means, clamp fractions), not the image.  Needs /root/reference; skipped when it is absent.
2. oracle_rays_<config>.npz -- ray batches (camera rays and the rays entering bounces 1 and 3 of seeded paths) with
   the oracle's Hittable::hit results, for every BASELINE config at test resolution.  These pin the oracle against
   silent change and give the GPU tests fixed inputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")


def reference_image_stats():
    path = "/root/reference/images/cornell_box_two_cylinders_medium_quality.png"
    if not os.path.exists(path):
        print("reference image not present; keeping the committed fixture")
        return
    from PIL import Image
    im = np.asarray(Image.open(path).convert("RGB")).astype(np.float64)
    assert im.shape == (600, 600, 3)
    blocks = im.reshape(6, 100, 6, 100, 3).mean(axis=(1, 3))
    np.savez(os.path.join(GOLD, "reference_cylinders_blockmeans.npz"), blocks=blocks, mean=im.mean(axis=(0, 1)),
             frac255=(im == 255).mean(), frac0=(im == 0).mean(),
             mean_linear=((im / 256.0) ** 2).mean(axis=(0, 1)))
    print("reference image: mean", im.mean(axis=(0, 1)), "frac255", (im == 255).mean())


def reference_models_image_stats():
    """images/cornell_models_medium_quality.png: cornell_model at 600x600 (200 spp, depth 50) with the author's own
    pawn and teapot meshes, which the reference's repository does not hold.  Its box, light and camera are C4's, so
    the blocks the models do not cover (ceiling, light, upper walls) pin C4's empty-box radiance coarsely."""
    path = "/root/reference/images/cornell_models_medium_quality.png"
    if not os.path.exists(path):
        print("reference models image not present; keeping the committed fixture")
        return
    from PIL import Image
    im = np.asarray(Image.open(path).convert("RGB")).astype(np.float64)
    assert im.shape == (600, 600, 3)
    np.savez(os.path.join(GOLD, "reference_models_blockmeans.npz"), blocks=im.reshape(6, 100, 6, 100, 3).mean(axis=(1, 3)),
             mean=im.mean(axis=(0, 1)), frac255=(im == 255).mean())
    print("reference models image: mean", im.mean(axis=(0, 1)))


def oracle_rays():
    from helpers import SMALL, build_pair, cameras, path_rays
    for cfg in ["C1", "C2", "C3", "C4", "C5"]:
        hs, os_ = build_pair(cfg, **SMALL[cfg])
        _, oc = cameras(hs)
        out = {}
        for bounce in (0, 1, 3):
            o, d, rng3 = path_rays(os_, oc, 700, bounce)
            o, d, rng3 = o[:512], d[:512], rng3[:512]
            r = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
            out[f"o{bounce}"], out[f"d{bounce}"], out[f"rng{bounce}"] = o, d, rng3
            for k in ("hit", "prim", "t", "p", "n", "uv", "front", "ctr"):
                out[f"{k}{bounce}"] = r[k]
        np.savez_compressed(os.path.join(GOLD, f"oracle_rays_{cfg}.npz"), **out)
        print(cfg, "rays written")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    reference_image_stats()
    reference_models_image_stats()
    if "--images-only" not in sys.argv:
        oracle_rays()
