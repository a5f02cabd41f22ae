"""ctypes front end of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, never
by the raytracer_amd package.  OracleBuilder has the same methods as raytracer_amd.builder.HostBuilder so that one
scene function (raytracer_amd/scenes.py) can describe the same scene to both."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librt_oracle.so")


class orc_camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("pixel00", C.c_double * 3), ("du", C.c_double * 3),
                ("dv", C.c_double * 3), ("center", C.c_double * 3), ("defocus_angle", C.c_double),
                ("disk_u", C.c_double * 3), ("disk_v", C.c_double * 3)]


class orc_render_opts(C.Structure):
    _fields_ = [("spp_begin", C.c_uint32), ("spp_count", C.c_uint32), ("max_depth", C.c_uint32),
                ("threads", C.c_uint32), ("seed", C.c_uint64), ("background", C.c_double * 3)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    vp, u32, u64, dbl, i32, cp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double, C.c_int, C.c_char_p
    L.orc_scene_new.restype = vp
    L.orc_scene_free.argtypes = [vp]
    L.orc_scene_free.restype = None
    L.orc_scene_error.restype = cp
    L.orc_scene_error.argtypes = [vp]
    sig = {
        "orc_tex_solid": [vp, dbl, dbl, dbl], "orc_tex_checker": [vp, dbl, i32, i32], "orc_tex_image": [vp, vp, u32, u32],
        "orc_tex_perlin": [vp, dbl, vp], "orc_mat_lambertian": [vp, i32], "orc_mat_metal": [vp, dbl, dbl, dbl, dbl],
        "orc_mat_dielectric": [vp, dbl], "orc_mat_diffuse_light": [vp, i32], "orc_mat_isotropic": [vp, i32],
        "orc_sphere": [vp, vp, dbl, i32], "orc_quad": [vp, vp, vp, vp, i32], "orc_triangle": [vp, vp, vp, vp, i32],
        "orc_y_cylinder": [vp, vp, dbl, i32], "orc_quadric_sphere": [vp, vp, dbl, i32], "orc_y_cone": [vp, vp, vp, i32],
        "orc_list_new": [vp], "orc_list_add": [vp, i32, i32], "orc_box": [vp, vp, vp, i32], "orc_slice": [vp, i32],
        "orc_bvh": [vp, i32], "orc_translate": [vp, i32, vp], "orc_rotate_y": [vp, i32, dbl],
        "orc_constant_medium": [vp, i32, dbl, i32], "orc_load_model": [vp, cp, dbl, i32], "orc_set_world": [vp, i32],
        "orc_bvh_depth": [vp, i32], "orc_bbox": [vp, i32, vp],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = i32
    L.orc_num_prims.restype = u32
    L.orc_num_prims.argtypes = [vp]
    L.orc_counts.restype = None
    L.orc_counts.argtypes = [vp, vp]
    L.orc_camera_new.restype = None
    L.orc_camera_new.argtypes = [dbl, u32, dbl, dbl, dbl, vp, vp, vp, C.POINTER(orc_camera)]
    L.orc_intersect_batch.restype = i32
    L.orc_intersect_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_trace_samples.restype = i32
    L.orc_trace_samples.argtypes = [vp, C.POINTER(orc_camera), C.POINTER(orc_render_opts), u64, vp, vp, vp, vp, i32, vp,
                                    vp, vp, vp]
    L.orc_render.restype = i32
    L.orc_render.argtypes = [vp, C.POINTER(orc_camera), C.POINTER(orc_render_opts), vp, vp]
    L.orc_tonemap_rgb8.restype = None
    L.orc_tonemap_rgb8.argtypes = [vp, u32, u32, u32, vp]
    L.orc_sphere_uv.restype = None
    L.orc_sphere_uv.argtypes = [vp, vp]
    L.orc_philox.restype = None
    L.orc_philox.argtypes = [vp, vp, vp]
    L.orc_rng_draw.restype = dbl
    L.orc_rng_draw.argtypes = [u64, u32, u32, u32, u32]
    L.orc_vec_op.restype = None
    L.orc_vec_op.argtypes = [i32, vp, vp, vp]
    L.orc_dot.restype = dbl
    L.orc_dot.argtypes = [vp, vp]
    L.orc_refract.restype = None
    L.orc_refract.argtypes = [vp, vp, dbl, vp]
    L.orc_reflectance.restype = dbl
    L.orc_reflectance.argtypes = [dbl, dbl]
    L.orc_process_color.restype = None
    L.orc_process_color.argtypes = [vp, vp]
    L.orc_aabb_hit.restype = i32
    L.orc_aabb_hit.argtypes = [vp, vp, vp, vp]
    L.orc_tex_value.restype = i32
    L.orc_tex_value.argtypes = [vp, i32, dbl, dbl, vp, vp]
    L.orc_mat_scatter.restype = i32
    L.orc_mat_scatter.argtypes = [vp, i32, u64, u32, u32, u32, vp, vp, vp, vp, dbl, dbl, i32, vp, vp, vp, vp, vp]
    _lib = L
    return L


def _d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


class OracleError(RuntimeError):
    pass


class OracleBuilder:
    def __init__(self):
        self.L = lib()
        self.s = self.L.orc_scene_new()

    def close(self):
        if self.s:
            self.L.orc_scene_free(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise OracleError(self.L.orc_scene_error(self.s).decode() or "bad handle")
        return rc

    def tex_solid(self, r, g, b):
        return self._ck(self.L.orc_tex_solid(self.s, r, g, b))

    def tex_checker(self, scale, even, odd):
        return self._ck(self.L.orc_tex_checker(self.s, scale, even, odd))

    def tex_image(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, _ = rgb.shape
        return self._ck(self.L.orc_tex_image(self.s, rgb.ctypes.data, w, h))

    def tex_perlin(self, scale, draws768):
        a = np.ascontiguousarray(draws768, dtype=np.float64)
        assert a.size == 768
        return self._ck(self.L.orc_tex_perlin(self.s, scale, a.ctypes.data))

    def mat_lambertian(self, tex):
        return self._ck(self.L.orc_mat_lambertian(self.s, tex))

    def mat_metal(self, r, g, b, fuzz):
        return self._ck(self.L.orc_mat_metal(self.s, r, g, b, fuzz))

    def mat_dielectric(self, ri):
        return self._ck(self.L.orc_mat_dielectric(self.s, ri))

    def mat_diffuse_light(self, tex):
        return self._ck(self.L.orc_mat_diffuse_light(self.s, tex))

    def mat_isotropic(self, tex):
        return self._ck(self.L.orc_mat_isotropic(self.s, tex))

    def sphere(self, c, r, mat):
        return self._ck(self.L.orc_sphere(self.s, _d3(c), r, mat))

    def quad(self, q, u, v, mat):
        return self._ck(self.L.orc_quad(self.s, _d3(q), _d3(u), _d3(v), mat))

    def triangle(self, q, u, v, mat):
        return self._ck(self.L.orc_triangle(self.s, _d3(q), _d3(u), _d3(v), mat))

    def y_cylinder(self, c, r, mat):
        return self._ck(self.L.orc_y_cylinder(self.s, _d3(c), r, mat))

    def quadric_sphere(self, c, r, mat):
        return self._ck(self.L.orc_quadric_sphere(self.s, _d3(c), r, mat))

    def y_cone(self, c, off, mat):
        return self._ck(self.L.orc_y_cone(self.s, _d3(c), _d3(off), mat))

    def list_new(self):
        return self._ck(self.L.orc_list_new(self.s))

    def list_add(self, lst, obj):
        self._ck(self.L.orc_list_add(self.s, lst, obj))

    def box(self, a, b, mat):
        return self._ck(self.L.orc_box(self.s, _d3(a), _d3(b), mat))

    def slice(self, lst):
        return self._ck(self.L.orc_slice(self.s, lst))

    def slice_of(self, objects):
        lst = self.list_new()
        for o in objects:
            self.list_add(lst, o)
        return self.slice(lst)

    def bvh(self, lst):
        return self._ck(self.L.orc_bvh(self.s, lst))

    def translate(self, obj, off):
        return self._ck(self.L.orc_translate(self.s, obj, _d3(off)))

    def rotate_y(self, obj, deg):
        return self._ck(self.L.orc_rotate_y(self.s, obj, deg))

    def constant_medium(self, boundary, density, tex):
        return self._ck(self.L.orc_constant_medium(self.s, boundary, density, tex))

    def load_model(self, path, scale, mat):
        return self._ck(self.L.orc_load_model(self.s, str(path).encode(), scale, mat))

    def bbox(self, obj):
        out = (C.c_double * 6)()
        self._ck(self.L.orc_bbox(self.s, obj, out))
        return np.array(out[:])

    def bvh_depth(self, obj):
        return self.L.orc_bvh_depth(self.s, obj)

    def set_world(self, obj):
        self._ck(self.L.orc_set_world(self.s, obj))

    def num_prims(self):
        return int(self.L.orc_num_prims(self.s))

    def counts(self):
        c = (C.c_uint64 * 8)()
        self.L.orc_counts(self.s, c)
        return dict(zip(["spheres", "quads", "triangles", "quadrics", "media", "bvh_nodes", "translates", "rotates"], c[:]))

    # ---- evaluation ----
    def intersect(self, o, d, tmin, tmax, rng3=None, seed=42):
        o = np.ascontiguousarray(o, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, dtype=np.float64), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, dtype=np.float64), (n,)))
        if rng3 is not None:
            rng3 = np.ascontiguousarray(rng3, dtype=np.uint32).reshape(n, 3)
        out = {"hit": np.zeros(n, np.uint8), "prim": np.zeros(n, np.uint32), "t": np.zeros(n), "p": np.zeros((n, 3)),
               "n": np.zeros((n, 3)), "uv": np.zeros((n, 2)), "front": np.zeros(n, np.uint8),
               "ctr": np.zeros(n, np.uint32)}
        cnt = (C.c_uint64 * 3)()
        rc = self.L.orc_intersect_batch(self.s, seed, o.ctypes.data, d.ctypes.data, tmin.ctypes.data, tmax.ctypes.data,
                                        rng3.ctypes.data if rng3 is not None else None, n, out["hit"].ctypes.data,
                                        out["prim"].ctypes.data, out["t"].ctypes.data, out["p"].ctypes.data,
                                        out["n"].ctypes.data, out["uv"].ctypes.data, out["front"].ctypes.data,
                                        out["ctr"].ctypes.data, cnt)
        if rc != 0:
            raise OracleError("no world set")
        out["counters"] = {"rays": cnt[0], "node_tests": cnt[1], "prim_tests": cnt[2]}
        return out

    def _opts(self, spp_begin, spp_count, max_depth, seed, background, threads=1):
        o = orc_render_opts()
        o.spp_begin, o.spp_count, o.max_depth, o.threads, o.seed = spp_begin, spp_count, max_depth, threads, seed
        o.background[0], o.background[1], o.background[2] = [float(x) for x in background]
        return o

    def trace_samples(self, cam, pixel, sample, max_depth, background, seed=42, capture_bounce=-1):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        n = pixel.size
        rad, nr = np.zeros((n, 3)), np.zeros(n, np.uint32)
        co, cd = np.zeros((n, 3)), np.zeros((n, 3))
        cv, cc = np.zeros(n, np.uint8), np.zeros(n, np.uint32)
        o = self._opts(0, 1, max_depth, seed, background)
        rc = self.L.orc_trace_samples(self.s, C.byref(cam), C.byref(o), n, pixel.ctypes.data, sample.ctypes.data,
                                      rad.ctypes.data, nr.ctypes.data, capture_bounce, co.ctypes.data, cd.ctypes.data,
                                      cv.ctypes.data, cc.ctypes.data)
        if rc != 0:
            raise OracleError("no world set")
        if capture_bounce >= 0:
            return rad, nr, {"o": co, "d": cd, "valid": cv, "ctr": cc}
        return rad, nr

    def render(self, cam, spp, max_depth, background, seed=42, spp_begin=0, threads=None):
        if threads is None:
            threads = os.cpu_count() or 1
        film = np.zeros((cam.height, cam.width, 3))
        cnt = (C.c_uint64 * 3)()
        o = self._opts(spp_begin, spp, max_depth, seed, background, threads)
        rc = self.L.orc_render(self.s, C.byref(cam), C.byref(o), film.ctypes.data, cnt)
        if rc != 0:
            raise OracleError("no world set")
        return film, {"rays": cnt[0], "node_tests": cnt[1], "prim_tests": cnt[2], "paths": cam.width * cam.height * spp}


def camera_new(aspect_ratio, image_width, vfov, defocus_angle, focus_distance, look_from, look_at, view_up):
    cam = orc_camera()
    lib().orc_camera_new(aspect_ratio, image_width, vfov, defocus_angle, focus_distance, _d3(look_from), _d3(look_at),
                         _d3(view_up), C.byref(cam))
    return cam


def tonemap(film, spp):
    film = np.ascontiguousarray(film, dtype=np.float64)
    h, w, _ = film.shape
    rgb = np.zeros((h, w, 3), np.uint8)
    lib().orc_tonemap_rgb8(film.ctypes.data, w, h, spp, rgb.ctypes.data)
    return rgb
