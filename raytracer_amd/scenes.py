"""The reference's scene programs (the five benchmark configurations first), written against the builder interface (raytracer_amd.builder.HostBuilder, or any
object with the same methods).  Each function follows one scene program of the reference's src/main.rs: same
objects, same order of `world.add` (order decides ties and the order of random draws), same camera.  Scene
randomness comes from philox.SceneRng in the reference's draw order, so `seed` names one scene.

Returns a SceneSpec: the builder with its world set, the camera parameters, and the config's render settings."""
import os
from dataclasses import dataclass, field

import numpy as np

from .philox import SceneRng

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


@dataclass
class SceneSpec:
    name: str
    builder: object
    camera: dict  # arguments of Camera::new
    width: int
    spp: int
    max_depth: int
    background: tuple
    info: dict = field(default_factory=dict)


def _lambertian(b, color):  # Lambertian::from_color, material.rs:71-75
    return b.mat_lambertian(b.tex_solid(*color))


def _light(b, color):  # DiffuseLight::from_color, material.rs:178-182
    return b.mat_diffuse_light(b.tex_solid(*color))


def _sub(a, c):
    return (a[0] - c[0], a[1] - c[1], a[2] - c[2])


def _len2(a):
    return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]


def many_spheres(b, seed=42, width=400, spp=10):
    """main.rs:40-153 (config 1).  488 spheres in one BVH, thin-lens camera."""
    rng = SceneRng(seed)
    world = b.list_new()
    checker = b.tex_checker(3.1, b.tex_solid(0.2, 0.3, 0.1), b.tex_solid(0.9, 0.9, 0.9))
    b.list_add(world, b.sphere((0.0, -1000.0, -1.0), 1000.0, b.mat_lambertian(checker)))
    glass_count = 0
    for a in range(-11, 11):
        for c in range(-11, 11):
            choose_mat = rng.range(0.0, 1.0)
            cx = float(a) + 0.9 * rng.range(0.0, 1.0)
            cz = float(c) + 0.9 * rng.range(0.0, 1.0)
            center = (cx, 0.2, cz)
            if _len2(_sub(center, (4.0, 0.2, 0.0))) > 0.0:
                if choose_mat < 0.8:
                    v1, v2 = rng.vector(0.0, 1.0), rng.vector(0.0, 1.0)
                    albedo = (v1[0] * v2[0], v1[1] * v2[1], v1[2] * v2[2])
                    b.list_add(world, b.sphere(center, 0.2, _lambertian(b, albedo)))
                elif choose_mat < 0.95:
                    v1, v2 = rng.vector(0.0, 1.0), rng.vector(0.0, 1.0)
                    albedo = (v1[0] * v2[0], v1[1] * v2[1], v1[2] * v2[2])
                    fuzz = rng.range(0.0, 0.5)
                    b.list_add(world, b.sphere(center, 0.2, b.mat_metal(albedo[0], albedo[1], albedo[2], fuzz)))
                else:
                    glass_count += 1
                    b.list_add(world, b.sphere(center, 0.2, b.mat_dielectric(1.5)))
    b.list_add(world, b.sphere((0.0, 1.0, 0.0), 1.0, b.mat_dielectric(1.5)))
    b.list_add(world, b.sphere((-4.0, 1.0, 0.0), 1.0, _lambertian(b, (0.4, 0.2, 0.1))))
    b.list_add(world, b.sphere((4.0, 1.0, 0.0), 1.0, b.mat_metal(0.7, 0.6, 0.5, 0.0)))
    b.set_world(b.bvh(world))  # BVHNode::from_hittable_list(world), main.rs:151
    cam = dict(aspect_ratio=16.0 / 9.0, vfov=20.0, defocus_angle=0.6, focus_distance=10.0,
               look_from=(13.0, 2.0, 3.0), look_at=(0.0, 0.0, 0.0), view_up=(0.0, 1.0, 0.0))
    return SceneSpec("many_spheres", b, cam, width, spp, 100, (0.7, 0.8, 1.0), {"glass": glass_count})


def _empty_cornell_box(b, world):
    """create_empty_cornell_box, main.rs:477-530."""
    red = _lambertian(b, (0.65, 0.05, 0.05))
    white = _lambertian(b, (0.73, 0.73, 0.73))
    green = _lambertian(b, (0.12, 0.45, 0.15))
    light = _light(b, (15.0, 15.0, 15.0))
    b.list_add(world, b.quad((555.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), green))
    b.list_add(world, b.quad((0.0, 0.0, 0.0), (0.0, 555.0, 0.0), (0.0, 0.0, 555.0), red))
    b.list_add(world, b.quad((113.0, 554.0, 127.0), (330.0, 0.0, 0.0), (0.0, 0.0, 305.0), light))
    b.list_add(world, b.quad((0.0, 555.0, 0.0), (555.0, 0.0, 0.0), (0.0, 0.0, 555.0), white))
    b.list_add(world, b.quad((0.0, 0.0, 0.0), (555.0, 0.0, 0.0), (0.0, 0.0, 555.0), white))
    b.list_add(world, b.quad((0.0, 0.0, 555.0), (555.0, 0.0, 0.0), (0.0, 555.0, 0.0), white))


_CORNELL_CAM = dict(aspect_ratio=1.0, vfov=40.0, defocus_angle=0.0, focus_distance=10.0,
                    look_from=(278.0, 278.0, -800.0), look_at=(278.0, 278.0, 0.0), view_up=(0.0, 1.0, 0.0))


def cornell_smoke(b, seed=42, width=600, spp=200):
    """main.rs:599-671 (config 2).  Two ConstantMedium boxes in the Cornell box."""
    world = b.list_new()
    _empty_cornell_box(b, world)
    white = _lambertian(b, (0.73, 0.73, 0.73))
    box1 = b.slice(b.box((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), white))
    box1 = b.translate(b.rotate_y(box1, 15.0), (265.0, 0.0, 295.0))
    box2 = b.slice(b.box((0.0, 0.0, 0.0), (165.0, 165.0, 165.0), white))
    box2 = b.translate(b.rotate_y(box2, -18.0), (130.0, 0.0, 65.0))
    b.list_add(world, b.constant_medium(box1, 0.01, b.tex_solid(0.0, 0.0, 0.0)))
    b.list_add(world, b.constant_medium(box2, 0.01, b.tex_solid(1.0, 1.0, 1.0)))
    b.set_world(b.slice(world))
    return SceneSpec("cornell_smoke", b, dict(_CORNELL_CAM), width, spp, 50, (0.0, 0.0, 0.0))


def cornell_quadric(b, seed=42, width=600, spp=1000):
    """main.rs:830-886 (config 3).  Two infinite y cylinders in the Cornell box."""
    world = b.list_new()
    _empty_cornell_box(b, world)
    white = _lambertian(b, (0.73, 0.73, 0.73))
    b.list_add(world, b.y_cylinder((150.0, 555.0 / 2.0, 175.0), 50.0, white))
    b.list_add(world, b.y_cylinder((400.0, 555.0 / 2.0, 555.0 / 2.0 + 50.0), 80.0, white))
    b.set_world(b.slice(world))
    return SceneSpec("cornell_quadric", b, dict(_CORNELL_CAM), width, spp, 50, (0.0, 0.0, 0.0))


def cornell_model(b, seed=42, width=800, spp=2000, pawn_path=None, teapot_path=None):
    """main.rs:1137-1198 (config 4).  Two .obj meshes, each in its own BVH, under Translate / RotateY."""
    world = b.list_new()
    _empty_cornell_box(b, world)
    white = _lambertian(b, (0.73, 0.73, 0.73))
    pawn = b.load_model(pawn_path or os.path.join(ASSETS, "pawn.obj"), 600.0, white)
    b.list_add(world, b.translate(pawn, (400.0, 200.0, 400.0)))
    teapot = b.load_model(teapot_path or os.path.join(ASSETS, "teapot.obj"), 50.0, white)
    b.list_add(world, b.rotate_y(b.translate(teapot, (75.0, 0.0, 200.0)), 50.0))
    b.set_world(b.slice(world))
    return SceneSpec("cornell_model", b, dict(_CORNELL_CAM), width, spp, 50, (0.0, 0.0, 0.0))


def load_ppm(path):
    with open(path, "rb") as fh:
        data = fh.read()
    parts = data.split(None, 4)
    if parts[0] != b"P6" or int(parts[3]) != 255:
        raise ValueError("not a binary 8-bit PPM: " + path)
    w, h = int(parts[1]), int(parts[2])
    return np.frombuffer(parts[4][:3 * w * h], dtype=np.uint8).reshape(h, w, 3)


def _earth_texture(b, earth_path=None):
    try:
        return b.tex_image(load_ppm(earth_path or os.path.join(ASSETS, "earthmap.ppm")))
    except (OSError, ValueError):
        # ImageTexture::new_or_fallback, texture.rs:94-111: a missing image becomes a magenta checker
        return b.tex_checker(2.0, b.tex_solid(1.0, 0.0, 0.862745098039),
                             b.tex_solid(0.00392156862745, 0.0, 0.00392156862745))


def final_scene(b, seed=42, width=800, spp=10000, max_depth=40, earth_path=None):
    """main.rs:673-828 (config 5 = final_scene(800, 10_000, 40), main.rs:1386)."""
    rng = SceneRng(seed)
    boxes1 = b.list_new()
    ground = _lambertian(b, (0.48, 0.83, 0.53))
    for i in range(20):
        for j in range(20):
            w = 100.0
            x0 = -1000.0 + float(i) * w
            z0 = -1000.0 + float(j) * w
            y0 = 0.0
            x1 = x0 + w
            y1 = rng.range(1.0, 101.0)
            z1 = z0 + w
            b.list_add(boxes1, b.slice(b.box((x0, y0, z0), (x1, y1, z1), ground)))
    world = b.list_new()
    b.list_add(world, b.bvh(boxes1))
    b.list_add(world, b.quad((123.0, 554.0, 147.0), (300.0, 0.0, 0.0), (0.0, 0.0, 265.0), _light(b, (7.0, 7.0, 7.0))))
    b.list_add(world, b.sphere((400.0, 400.0, 200.0), 50.0, _lambertian(b, (0.7, 0.3, 0.1))))
    b.list_add(world, b.sphere((260.0, 150.0, 45.0), 50.0, b.mat_dielectric(1.5)))
    b.list_add(world, b.sphere((0.0, 150.0, 145.0), 50.0, b.mat_metal(0.8, 0.8, 0.9, 1.0)))
    boundary = b.sphere((360.0, 150.0, 145.0), 70.0, b.mat_dielectric(1.5))
    b.list_add(world, boundary)
    b.list_add(world, b.constant_medium(boundary, 0.2, b.tex_solid(0.2, 0.4, 0.9)))
    boundary2 = b.sphere((0.0, 0.0, 0.0), 5000.0, b.mat_dielectric(1.5))
    b.list_add(world, b.constant_medium(boundary2, 0.0001, b.tex_solid(1.0, 1.0, 1.0)))
    earth = _earth_texture(b, earth_path)
    b.list_add(world, b.sphere((400.0, 200.0, 400.0), 100.0, b.mat_lambertian(earth)))
    draws = [rng.range(-1.0, 1.0) for _ in range(768)]  # create_perlin_noise, perlin.rs:19
    perlin = b.tex_perlin(0.2, draws)
    b.list_add(world, b.sphere((220.0, 280.0, 300.0), 80.0, b.mat_lambertian(perlin)))
    spheres = b.list_new()
    white = _lambertian(b, (0.73, 0.73, 0.73))
    for _ in range(1000):
        b.list_add(spheres, b.sphere(rng.vector(0.0, 165.0), 10.0, white))
    b.list_add(world, b.translate(b.rotate_y(b.bvh(spheres), 15.0), (-100.0, 270.0, 395.0)))
    b.set_world(b.slice(world))
    cam = dict(aspect_ratio=1.0, vfov=40.0, defocus_angle=0.0, focus_distance=10.0,
               look_from=(478.0, 278.0, -600.0), look_at=(278.0, 278.0, 0.0), view_up=(0.0, 1.0, 0.0))
    return SceneSpec("final_scene", b, cam, width, spp, max_depth, (0.0, 0.0, 0.0))


# ---- the reference's other scene programs (main.rs:1373-1397 lists all of them) ----

_SKY = (0.7, 0.8, 1.0)
_WIDE_CAM = dict(aspect_ratio=16.0 / 9.0, vfov=20.0, defocus_angle=0.0, focus_distance=10.0,
                 look_from=(13.0, 2.0, 3.0), look_at=(0.0, 0.0, 0.0), view_up=(0.0, 1.0, 0.0))


def checkered_spheres(b, seed=42, width=400, spp=100):
    """main.rs:155-208.  Two large spheres sharing one checker texture."""
    world = b.list_new()
    checker = b.tex_checker(0.10, b.tex_solid(0.2, 0.3, 0.1), b.tex_solid(0.9, 0.9, 0.9))
    mat = b.mat_lambertian(checker)
    b.list_add(world, b.sphere((0.0, -10.0, 0.0), 10.0, mat))
    b.list_add(world, b.sphere((0.0, 10.0, 0.0), 10.0, mat))
    b.set_world(b.slice(world))
    return SceneSpec("checkered_spheres", b, dict(_WIDE_CAM), width, spp, 50, _SKY)


def earth(b, seed=42, width=400, spp=100, earth_path=None):
    """main.rs:210-259.  One sphere with the image texture (or its magenta fallback checker)."""
    world = b.list_new()
    b.list_add(world, b.sphere((0.0, 0.0, 0.0), 2.0, b.mat_lambertian(_earth_texture(b, earth_path))))
    b.set_world(b.slice(world))
    cam = dict(_WIDE_CAM, look_from=(0.0, 0.0, 12.0))
    return SceneSpec("earth", b, cam, width, spp, 50, _SKY)


def _perlin_pair(b, rng, world):
    draws = [rng.range(-1.0, 1.0) for _ in range(768)]  # create_perlin_noise, perlin.rs:19
    mat = b.mat_lambertian(b.tex_perlin(2.0, draws))
    b.list_add(world, b.sphere((0.0, -1000.0, 0.0), 1000.0, mat))
    b.list_add(world, b.sphere((0.0, 2.0, 0.0), 2.0, mat))


def perlin_spheres(b, seed=42, width=400, spp=100):
    """main.rs:261-320.  Ground and one sphere with the turbulence texture."""
    world = b.list_new()
    _perlin_pair(b, SceneRng(seed), world)
    b.set_world(b.slice(world))
    return SceneSpec("perlin_spheres", b, dict(_WIDE_CAM), width, spp, 50, _SKY)


def para(b, seed=42, width=400, spp=100):
    """main.rs:322-402.  Five parallelograms facing the camera."""
    world = b.list_new()
    b.list_add(world, b.quad((-3.0, -2.0, 5.0), (0.0, 0.0, -4.0), (0.0, 4.0, 0.0), _lambertian(b, (1.0, 0.2, 0.2))))
    b.list_add(world, b.quad((-2.0, -2.0, 0.0), (4.0, 0.0, 0.0), (0.0, 4.0, 0.0), _lambertian(b, (0.2, 1.0, 0.2))))
    b.list_add(world, b.quad((3.0, -2.0, 1.0), (0.0, 0.0, 4.0), (0.0, 4.0, 0.0), _lambertian(b, (0.2, 0.2, 1.0))))
    b.list_add(world, b.quad((-2.0, 3.0, 1.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), _lambertian(b, (1.0, 0.5, 0.0))))
    b.list_add(world, b.quad((-2.0, -3.0, 5.0), (4.0, 0.0, 0.0), (0.0, 0.0, -4.0), _lambertian(b, (0.2, 0.8, 0.8))))
    b.set_world(b.slice(world))
    cam = dict(aspect_ratio=1.0, vfov=80.0, defocus_angle=0.0, focus_distance=10.0,
               look_from=(0.0, 0.0, 9.0), look_at=(0.0, 0.0, 0.0), view_up=(0.0, 1.0, 0.0))
    return SceneSpec("para", b, cam, width, spp, 50, _SKY)


def simple_light(b, seed=42, width=400, spp=100):
    """main.rs:404-475.  The perlin pair lit by one parallelogram against a black background."""
    world = b.list_new()
    _perlin_pair(b, SceneRng(seed), world)
    b.list_add(world, b.quad((3.0, 1.0, -2.0), (2.0, 0.0, 0.0), (0.0, 2.0, 0.0), _light(b, (4.0, 4.0, 4.0))))
    b.set_world(b.slice(world))
    cam = dict(_WIDE_CAM, look_from=(26.0, 3.0, 6.0), look_at=(0.0, 2.0, 0.0))
    return SceneSpec("simple_light", b, cam, width, spp, 50, (0.0, 0.0, 0.0))


def cornell_box(b, seed=42, width=600, spp=20):
    """main.rs:532-597.  The two solid boxes under RotateY / Translate."""
    world = b.list_new()
    _empty_cornell_box(b, world)
    white = _lambertian(b, (0.73, 0.73, 0.73))
    box1 = b.slice(b.box((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), white))
    b.list_add(world, b.translate(b.rotate_y(box1, 15.0), (265.0, 0.0, 295.0)))
    box2 = b.slice(b.box((0.0, 0.0, 0.0), (165.0, 165.0, 165.0), white))
    b.list_add(world, b.translate(b.rotate_y(box2, -18.0), (130.0, 0.0, 65.0)))
    b.set_world(b.slice(world))
    return SceneSpec("cornell_box", b, dict(_CORNELL_CAM), width, spp, 50, (0.0, 0.0, 0.0))


def debug_quadric(b, seed=42, width=600, spp=50):
    """main.rs:888-946.  A sphere and an infinite cylinder under one light."""
    world = b.list_new()
    b.list_add(world, b.quad((10.0, 10.0, 10.0), (10.0, 0.0, 10.0), (0.0, 10.0, 10.0), _light(b, (15.0, 15.0, 15.0))))
    white = _lambertian(b, (0.73, 0.73, 0.73))
    b.list_add(world, b.sphere((3.0, 0.0, 0.0), 1.0, white))
    b.list_add(world, b.y_cylinder((0.0, 0.0, 0.0), 1.0, white))
    b.set_world(b.slice(world))
    cam = dict(aspect_ratio=1.0, vfov=40.0, defocus_angle=0.0, focus_distance=10.0,
               look_from=(0.0, 0.0, 12.0), look_at=(0.0, 0.0, 0.0), view_up=(0.0, 1.0, 0.0))
    return SceneSpec("debug_quadric", b, cam, width, spp, 50, (0.0, 0.0, 0.0))


def cornell_triangle(b, seed=42, width=300, spp=20):
    """main.rs:948-998.  One triangle in the Cornell box; ImageQuality::low() = 20 spp, depth 4 (camera.rs:64-69)."""
    world = b.list_new()
    _empty_cornell_box(b, world)
    white = _lambertian(b, (0.73, 0.73, 0.73))
    b.list_add(world, b.triangle((555.0 / 2.0, 555.0 / 2.0, 555.0 / 2.0), (100.0, 100.0, 10.0), (100.0, 0.0, 100.0), white))
    b.set_world(b.slice(world))
    return SceneSpec("cornell_triangle", b, dict(_CORNELL_CAM), width, spp, 4, (0.0, 0.0, 0.0))


def profiler_scene(b, seed=42, width=400, spp=20, max_depth=4, earth_path=None):
    """main.rs:1000-1135 (called as profiler_scene(400, 20, 4), main.rs:1390): final_scene without the global
    fog and the perlin sphere."""
    rng = SceneRng(seed)
    boxes1 = b.list_new()
    ground = _lambertian(b, (0.48, 0.83, 0.53))
    for i in range(20):
        for j in range(20):
            w = 100.0
            x0 = -1000.0 + float(i) * w
            z0 = -1000.0 + float(j) * w
            y1 = rng.range(1.0, 101.0)
            b.list_add(boxes1, b.slice(b.box((x0, 0.0, z0), (x0 + w, y1, z0 + w), ground)))
    world = b.list_new()
    b.list_add(world, b.bvh(boxes1))
    b.list_add(world, b.quad((123.0, 554.0, 147.0), (300.0, 0.0, 0.0), (0.0, 0.0, 265.0), _light(b, (7.0, 7.0, 7.0))))
    b.list_add(world, b.sphere((400.0, 400.0, 200.0), 50.0, _lambertian(b, (0.7, 0.3, 0.1))))
    b.list_add(world, b.sphere((260.0, 150.0, 45.0), 50.0, b.mat_dielectric(1.5)))
    b.list_add(world, b.sphere((0.0, 150.0, 145.0), 50.0, b.mat_metal(0.8, 0.8, 0.9, 1.0)))
    boundary = b.sphere((360.0, 150.0, 145.0), 70.0, b.mat_dielectric(1.5))
    b.list_add(world, boundary)
    b.list_add(world, b.constant_medium(boundary, 0.2, b.tex_solid(0.2, 0.4, 0.9)))
    b.list_add(world, b.sphere((400.0, 200.0, 400.0), 100.0, b.mat_lambertian(_earth_texture(b, earth_path))))
    spheres = b.list_new()
    white = _lambertian(b, (0.73, 0.73, 0.73))
    for _ in range(1000):
        b.list_add(spheres, b.sphere(rng.vector(0.0, 165.0), 10.0, white))
    b.list_add(world, b.translate(b.rotate_y(b.bvh(spheres), 15.0), (-100.0, 270.0, 395.0)))
    b.set_world(b.slice(world))
    cam = dict(aspect_ratio=1.0, vfov=40.0, defocus_angle=0.0, focus_distance=10.0,
               look_from=(478.0, 278.0, -600.0), look_at=(278.0, 278.0, 0.0), view_up=(0.0, 1.0, 0.0))
    return SceneSpec("profiler_scene", b, cam, width, spp, max_depth, (0.0, 0.0, 0.0))


def spherical_mirror(b, seed=42, width=800, spp=800):
    """main.rs:1200-1277.  A sphere light, a mirror quadric sphere, 100 small spheres in a BVH, a floor."""
    rng = SceneRng(seed)
    world = b.list_new()
    b.list_add(world, b.sphere((0.0, 300.0, 0.0), 100.0, _light(b, (15.0, 15.0, 15.0))))
    b.list_add(world, b.quadric_sphere((0.0, 0.0, 0.0), 10.0, b.mat_metal(0.8, 0.8, 0.8, 0.0)))
    spheres = b.list_new()
    white = _lambertian(b, (0.73, 0.73, 0.73))
    for _ in range(100):
        b.list_add(spheres, b.sphere(rng.vector(-25.0, 25.0), 1.0, white))
    b.list_add(world, b.bvh(spheres))
    blue = _lambertian(b, (0.0, 48.0 / 255.0, 143.0 / 255.0))
    b.list_add(world, b.quad((-30.0, -30.0, -30.0), (60.0, 0.0, 0.0), (0.0, 0.0, 60.0), blue))
    b.set_world(b.slice(world))
    cam = dict(aspect_ratio=1.0, vfov=40.0, defocus_angle=0.0, focus_distance=10.0,
               look_from=(-100.0, 0.0, 0.0), look_at=(0.0, 0.0, 0.0), view_up=(0.0, 1.0, 0.0))
    return SceneSpec("spherical_mirror", b, cam, width, spp, 50, (0.0, 0.0, 0.0))


def _model_scene(name, b, path, look_from, width, spp):
    world = b.list_new()
    white = _lambertian(b, (0.73, 0.73, 0.73))
    b.list_add(world, b.load_model(path, 1.0, white))
    b.set_world(b.slice(world))
    cam = dict(aspect_ratio=16.0 / 9.0, vfov=40.0, defocus_angle=0.0, focus_distance=10.0,
               look_from=look_from, look_at=(0.0, 0.0, 0.0), view_up=(0.0, 1.0, 0.0))
    return SceneSpec(name, b, cam, width, spp, 4, (1.0, 1.0, 1.0))


def debug_model(b, seed=42, width=600, spp=20, model_path=None):
    """main.rs:1279-1324.  One .obj mesh at scale 1 against a white background, ImageQuality::low()."""
    return _model_scene("debug_model", b, model_path or os.path.join(ASSETS, "pawn.obj"), (0.0, 0.0, 2.0), width, spp)


def bust(b, seed=42, width=600, spp=20, model_path=None, triangles=766_484):
    """main.rs:1326-1371.  As debug_model from 200 units away.  The reference's meshes (models/Emperor.obj, or
    models/David.obj with 766 484 triangles, main.rs:1330-1331) are not part of its repository: unless a path is
    given, a procedural mesh of `triangles` triangles (raytracer_amd.meshgen) is written to a temporary .obj and goes
    through load_model like any other file."""
    if model_path is None:
        from .meshgen import cached_bust
        model_path = cached_bust(triangles)
    return _model_scene("bust", b, model_path, (0.0, 0.0, 200.0), width, spp)


# BASELINE.json configs, in order
CONFIGS = {
    "C1": many_spheres,
    "C2": cornell_smoke,
    "C3": cornell_quadric,
    "C4": cornell_model,
    "C5": final_scene,
}


# every scene program of main.rs by its function name; the match in main() (main.rs:1377-1397) numbers them
SCENES = {f.__name__: f for f in (
    many_spheres, checkered_spheres, earth, perlin_spheres, para, simple_light, cornell_box, cornell_smoke,
    final_scene, cornell_quadric, debug_quadric, cornell_triangle, profiler_scene, cornell_model, spherical_mirror,
    debug_model, bust)}


def build(config, builder, **kw):
    """`config` is a BASELINE key (C1..C5) or a scene program's name."""
    return (CONFIGS.get(config) or SCENES[config])(builder, **kw)
