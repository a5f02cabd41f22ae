"""Scene builder over the C++ host layer (librt_host.so).  One method per constructor of the reference's
scene-building API; see include/rt_host.h for the reference citations."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RtError


def _d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


class HostBuilder:
    def __init__(self):
        self.H = _lib.host()
        self.b = self.H.rth_new()
        if not self.b:
            raise RtError("rth_new failed")
        self._keep = []

    def close(self):
        if self.b:
            self.H.rth_free(self.b)
            self.b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise RtError(self.H.rth_error(self.b).decode())
        return rc

    # textures
    def tex_solid(self, r, g, b):
        return self._ck(self.H.rth_tex_solid(self.b, r, g, b))

    def tex_checker(self, scale, even, odd):
        return self._ck(self.H.rth_tex_checker(self.b, scale, even, odd))

    def tex_image(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, _ = rgb.shape
        return self._ck(self.H.rth_tex_image(self.b, rgb.ctypes.data, w, h))

    def tex_image_file(self, path):
        return self._ck(self.H.rth_tex_image_file(self.b, str(path).encode()))

    def tex_perlin(self, scale, draws768):
        a = np.ascontiguousarray(draws768, dtype=np.float64)
        assert a.size == 768
        return self._ck(self.H.rth_tex_perlin(self.b, scale, a.ctypes.data))

    # materials
    def mat_lambertian(self, tex):
        return self._ck(self.H.rth_mat_lambertian(self.b, tex))

    def mat_metal(self, r, g, b, fuzz):
        return self._ck(self.H.rth_mat_metal(self.b, r, g, b, fuzz))

    def mat_dielectric(self, ri):
        return self._ck(self.H.rth_mat_dielectric(self.b, ri))

    def mat_diffuse_light(self, tex):
        return self._ck(self.H.rth_mat_diffuse_light(self.b, tex))

    def mat_isotropic(self, tex):
        return self._ck(self.H.rth_mat_isotropic(self.b, tex))

    # primitives
    def sphere(self, c, r, mat):
        return self._ck(self.H.rth_sphere(self.b, _d3(c), r, mat))

    def quad(self, q, u, v, mat):
        return self._ck(self.H.rth_quad(self.b, _d3(q), _d3(u), _d3(v), mat))

    def triangle(self, q, u, v, mat):
        return self._ck(self.H.rth_triangle(self.b, _d3(q), _d3(u), _d3(v), mat))

    def y_cylinder(self, c, r, mat):
        return self._ck(self.H.rth_y_cylinder(self.b, _d3(c), r, mat))

    def quadric_sphere(self, c, r, mat):
        return self._ck(self.H.rth_quadric_sphere(self.b, _d3(c), r, mat))

    def y_cone(self, c, off, mat):
        return self._ck(self.H.rth_y_cone(self.b, _d3(c), _d3(off), mat))

    # collections and instances
    def list_new(self):
        return self._ck(self.H.rth_list_new(self.b))

    def list_add(self, lst, obj):
        self._ck(self.H.rth_list_add(self.b, lst, obj))

    def box(self, a, b, mat):
        return self._ck(self.H.rth_box(self.b, _d3(a), _d3(b), mat))

    def slice(self, lst):
        return self._ck(self.H.rth_slice(self.b, lst))

    def slice_of(self, objects):
        """to_hittable_slice of a fresh list holding `objects` in order."""
        lst = self.list_new()
        for o in objects:
            self.list_add(lst, o)
        return self.slice(lst)

    def bvh(self, lst):
        return self._ck(self.H.rth_bvh(self.b, lst))

    def translate(self, obj, off):
        return self._ck(self.H.rth_translate(self.b, obj, _d3(off)))

    def rotate_y(self, obj, deg):
        return self._ck(self.H.rth_rotate_y(self.b, obj, deg))

    def constant_medium(self, boundary, density, tex):
        return self._ck(self.H.rth_constant_medium(self.b, boundary, density, tex))

    def load_model(self, path, scale, mat):
        return self._ck(self.H.rth_load_model(self.b, str(path).encode(), scale, mat))

    def bbox(self, obj):
        out = (C.c_double * 6)()
        self._ck(self.H.rth_bbox(self.b, obj, out))
        return np.array(out[:])

    def extent(self, obj):
        """Geometric bounds of the object's possible hit points (x0 x1 y0 y1 z0 z1), the source of the cull boxes."""
        out = (C.c_double * 6)()
        self._ck(self.H.rth_extent(self.b, obj, out))
        return np.array(out[:])

    def set_world(self, obj):
        self._ck(self.H.rth_set_world(self.b, obj))
        self.world = obj

    # flattened view
    def num_prims(self):
        return int(self.H.rth_num_prim_ids(self.b))

    _FLAT = {"nodes": (0, _lib.rt_node), "boxes": (1, 6), "spheres": (2, 4), "quads": (3, 16), "triangles": (4, 16),
             "quadrics": (5, 10), "media": (6, _lib.rt_medium), "ops": (7, _lib.rt_xform_op),
             "chains": (8, _lib.rt_chain), "materials": (9, _lib.rt_material), "textures": (10, _lib.rt_texture),
             "sphere_id": (11, "u32"), "quad_id": (12, "u32"), "tri_id": (13, "u32"), "quadric_id": (14, "u32"),
             "sphere_mat": (15, "u32"), "quad_mat": (16, "u32"), "tri_mat": (17, "u32"), "quadric_mat": (18, "u32"),
             "cull_boxes": (23, 6)}

    def flat_ptr(self, what):
        """(address, count) of flattened array number `what` (see include/rt_host.h), owned by the builder."""
        n = C.c_uint64(0)
        p = self.H.rth_flat_array(self.b, what, C.byref(n))
        return (p or 0), n.value

    def flat(self, name):
        """Copy of one flattened array: numpy for numbers, a list of ctypes structs for records."""
        what, ty = self._FLAT[name]
        n = C.c_uint64(0)
        p = self.H.rth_flat_array(self.b, what, C.byref(n))
        n = n.value
        if isinstance(ty, int):
            if n == 0:
                return np.zeros((0, ty))
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n * ty,)).reshape(n, ty).copy()
        if ty == "u32":
            if n == 0:
                return np.zeros((0,), dtype=np.uint32)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n,)).copy()
        arr = C.cast(p, C.POINTER(ty))
        out = []
        for i in range(n):
            rec = ty()
            C.memmove(C.byref(rec), C.byref(arr[i]), C.sizeof(ty))
            out.append(rec)
        return out

    def thrender(self, world, cam, spp, max_depth, background, output_path, seed=42, device=0, print_metrics=False):
        """Camera::thrender of the host layer: renders the tree under `world` on the GPU and writes the PNG."""
        st = _lib.rt_stats()
        self._ck(self.H.rth_thrender(self.b, world, C.byref(cam), spp, max_depth, _d3(background), seed, device,
                                     str(output_path).encode(), 1 if print_metrics else 0, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _lib.rt_stats._fields_}

    def commit(self, device=0):
        """Uploads the flattened world to a GPU and returns a Scene."""
        from .scene import Scene
        h = self.H.rth_commit(self.b, device)
        if not h:
            raise RtError("commit failed: " + self.H.rth_error(self.b).decode())
        return Scene(h, device)


def camera_new(aspect_ratio, image_width, vfov, defocus_angle, focus_distance, look_from, look_at, view_up):
    """Camera::new (camera.rs:80-143), evaluated by the C++ host layer."""
    cam = _lib.rt_camera()
    _lib.host().rth_camera_new(aspect_ratio, image_width, vfov, defocus_angle, focus_distance, _d3(look_from),
                               _d3(look_at), _d3(view_up), C.byref(cam))
    return cam


def read_image(path):
    """The decoder behind ImageTexture::new_or_fallback: binary PPM or PNG -> (H, W, 3) uint8, or None."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    H = _lib.host()
    if H.rth_read_image(str(path).encode(), None, 0, C.byref(w), C.byref(h)) != 0:
        return None
    out = np.zeros((h.value, w.value, 3), np.uint8)
    if H.rth_read_image(str(path).encode(), out.ctypes.data, out.size, C.byref(w), C.byref(h)) != 0:
        return None
    return out


def write_png(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    if _lib.host().rth_write_png(str(path).encode(), rgb.ctypes.data, w, h) != 0:
        raise RtError(f"could not write {path}")
