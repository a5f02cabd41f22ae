"""ctypes bindings of the two shared libraries: librt_amd.so (HIP kernels + C ABI, include/rt_amd.h) and
librt_host.so (C++ scene-building layer, include/rt_host.h).  Plumbing only: no arithmetic happens here."""
import atexit
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")


class RtError(RuntimeError):
    pass


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("albedo", C.c_double * 3), ("param", C.c_double)]


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("reserved", C.c_uint32),
                ("color", C.c_double * 3), ("scale", C.c_double)]


class rt_xform_op(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reserved", C.c_uint32), ("v", C.c_double * 3)]


class rt_chain(C.Structure):
    _fields_ = [("first_op", C.c_uint32), ("n_ops", C.c_uint32)]


class rt_medium(C.Structure):
    _fields_ = [("neg_inv_density", C.c_double), ("material", C.c_uint32), ("prim_id", C.c_uint32)]


class rt_node(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("level", C.c_uint8), ("chain", C.c_uint16), ("skip", C.c_uint32),
                ("first", C.c_uint32), ("count", C.c_uint32)]


class rt_camera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("pixel00", C.c_double * 3),
                ("delta_u", C.c_double * 3), ("delta_v", C.c_double * 3), ("center", C.c_double * 3),
                ("defocus_angle", C.c_double), ("disk_u", C.c_double * 3), ("disk_v", C.c_double * 3)]


class rt_render_opts(C.Structure):
    _fields_ = [("spp_begin", C.c_uint32), ("spp_count", C.c_uint32), ("max_depth", C.c_uint32),
                ("flags", C.c_uint32), ("seed", C.c_uint64), ("background", C.c_double * 3),
                ("batch_paths", C.c_uint64)]


class rt_stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("node_tests", C.c_uint64), ("prim_tests", C.c_uint64),
                ("cull_tests", C.c_uint64), ("prim_bytes", C.c_uint64), ("prim_flops", C.c_uint64),
                ("ms_total", C.c_double), ("ms_traverse", C.c_double), ("ms_shade", C.c_double),
                ("ms_paths", C.c_double), ("ms_bounce", C.c_double), ("ms_finish", C.c_double), ("traverse_launches", C.c_uint32),
                ("path_launches", C.c_uint32), ("bounce_launches", C.c_uint32), ("bounces", C.c_uint32),
                ("lockstep", C.c_uint32), ("finish_launches", C.c_uint32), ("list_rays", C.c_uint64), ("wide_tests", C.c_uint64),
                ("shade_quads", C.c_uint64)]


RT_FLAG_COUNTERS = 1
RT_FLAG_NO_CULL = 2
RT_FLAG_WHOLE_PATHS = 4
RT_FLAG_QUEUES = 8
RT_FLAG_SPLIT_BOUNCE = 16
RT_FLAG_FAST_BVH = 32
RT_NO_HIT = 0xFFFFFFFF
NODE_KINDS = {0: "bvh", 1: "medium", 2: "spheres", 3: "quads", 4: "triangles", 5: "quadrics", 6: "cull"}

# every symbol include/rt_amd.h declares
RT_AMD_SYMBOLS = [
    "rt_scene_create", "rt_scene_destroy", "rt_last_error", "rt_scene_set_materials", "rt_scene_set_textures",
    "rt_scene_set_image", "rt_scene_set_perlin", "rt_scene_set_spheres", "rt_scene_set_quads",
    "rt_scene_set_triangles", "rt_scene_set_quadrics", "rt_scene_set_media", "rt_scene_set_xforms",
    "rt_scene_set_nodes", "rt_scene_set_cull_boxes", "rt_scene_commit", "rt_intersect_batch", "rt_render", "rt_render_device",
    "rt_render_multi", "rt_trace_samples", "rt_tonemap_rgb8", "rt_debug_arith", "rt_intersect_batch_noalloc",
    "rt_scene_set_fast_bvh", "rt_multi_rccl_reductions", "rt_debug_device_allocs", "rt_multi_shutdown",
]

_amd = None
_host = None


def _load(name):
    path = os.path.join(_CSRC, name)
    if not os.path.exists(path):
        raise RtError(f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()' or make)")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def amd():
    """librt_amd.so with argument types declared.  Raises if the library has not been built: there is no fallback."""
    global _amd
    if _amd is not None:
        return _amd
    L = _load("librt_amd.so")
    vp, u32, u64, dbl, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double, C.c_int
    L.rt_scene_create.restype = vp
    L.rt_scene_create.argtypes = []
    L.rt_scene_destroy.restype = None
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_last_error.argtypes = []
    L.rt_scene_set_materials.argtypes = [vp, vp, u32]
    L.rt_scene_set_textures.argtypes = [vp, vp, u32]
    L.rt_scene_set_image.argtypes = [vp, u32, vp, u32, u32]
    L.rt_scene_set_perlin.argtypes = [vp, u32, vp, vp]
    for f in (L.rt_scene_set_spheres, L.rt_scene_set_quads, L.rt_scene_set_triangles, L.rt_scene_set_quadrics):
        f.argtypes = [vp, vp, vp, vp, u32]
    L.rt_scene_set_media.argtypes = [vp, vp, u32]
    L.rt_scene_set_xforms.argtypes = [vp, vp, u32, vp, u32]
    L.rt_scene_set_nodes.argtypes = [vp, vp, u32, vp, u32]
    L.rt_scene_set_cull_boxes.argtypes = [vp, vp, u32]
    L.rt_scene_commit.argtypes = [vp, i32]
    L.rt_scene_set_fast_bvh.argtypes = [vp, i32]
    L.rt_intersect_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, u64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rt_intersect_batch_noalloc.argtypes = [vp, u64, vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
    L.rt_render.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts), vp, C.POINTER(rt_stats)]
    L.rt_render_device.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts), vp, C.POINTER(rt_stats)]
    L.rt_render_multi.argtypes = [vp, i32, C.POINTER(rt_camera), C.POINTER(rt_render_opts), vp, C.POINTER(rt_stats)]
    L.rt_trace_samples.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_render_opts), u64, vp, vp, vp, vp]
    L.rt_tonemap_rgb8.argtypes = [vp, vp, u32, u32, u32, vp]
    L.rt_debug_arith.argtypes = [i32, vp, vp, u64, vp, vp, vp, vp]
    for name in RT_AMD_SYMBOLS:
        f = getattr(L, name)
        if name not in ("rt_scene_create", "rt_scene_destroy", "rt_last_error", "rt_multi_rccl_reductions", "rt_debug_device_allocs", "rt_multi_shutdown"):
            f.restype = i32
    L.rt_multi_rccl_reductions.argtypes = []
    L.rt_multi_rccl_reductions.restype = u64
    L.rt_debug_device_allocs.argtypes = []
    L.rt_debug_device_allocs.restype = u64
    L.rt_multi_shutdown.argtypes = []
    L.rt_multi_shutdown.restype = None
    atexit.register(L.rt_multi_shutdown)  # (RCCL communicators go before the interpreter lets go of the HIP runtime)
    _amd = L
    return L


def host():
    """librt_host.so (links librt_amd.so)."""
    global _host
    if _host is not None:
        return _host
    amd()
    H = _load("librt_host.so")
    vp, u32, dbl, i32, cp = C.c_void_p, C.c_uint32, C.c_double, C.c_int, C.c_char_p
    H.rth_new.restype = vp
    H.rth_new.argtypes = []
    H.rth_free.restype = None
    H.rth_free.argtypes = [vp]
    H.rth_error.restype = cp
    H.rth_error.argtypes = [vp]
    sig = {
        "rth_tex_solid": [vp, dbl, dbl, dbl], "rth_tex_checker": [vp, dbl, i32, i32],
        "rth_tex_image": [vp, vp, u32, u32], "rth_tex_image_file": [vp, cp], "rth_tex_perlin": [vp, dbl, vp],
        "rth_mat_lambertian": [vp, i32], "rth_mat_metal": [vp, dbl, dbl, dbl, dbl], "rth_mat_dielectric": [vp, dbl],
        "rth_mat_diffuse_light": [vp, i32], "rth_mat_isotropic": [vp, i32],
        "rth_sphere": [vp, vp, dbl, i32], "rth_quad": [vp, vp, vp, vp, i32], "rth_triangle": [vp, vp, vp, vp, i32],
        "rth_y_cylinder": [vp, vp, dbl, i32], "rth_quadric_sphere": [vp, vp, dbl, i32], "rth_y_cone": [vp, vp, vp, i32],
        "rth_list_new": [vp], "rth_list_add": [vp, i32, i32], "rth_box": [vp, vp, vp, i32], "rth_slice": [vp, i32],
        "rth_bvh": [vp, i32], "rth_translate": [vp, i32, vp], "rth_rotate_y": [vp, i32, dbl],
        "rth_constant_medium": [vp, i32, dbl, i32], "rth_load_model": [vp, cp, dbl, i32], "rth_bbox": [vp, i32, vp], "rth_extent": [vp, i32, vp],
        "rth_set_world": [vp, i32],
    }
    for name, args in sig.items():
        f = getattr(H, name)
        f.argtypes = args
        f.restype = i32
    H.rth_num_prim_ids.restype = u32
    H.rth_num_prim_ids.argtypes = [vp]
    H.rth_flat_array.restype = vp
    H.rth_flat_array.argtypes = [vp, i32, C.POINTER(C.c_uint64)]
    H.rth_commit.restype = vp
    H.rth_commit.argtypes = [vp, i32]
    H.rth_camera_new.restype = None
    H.rth_camera_new.argtypes = [dbl, u32, dbl, dbl, dbl, vp, vp, vp, C.POINTER(rt_camera)]
    H.rth_thrender.restype = i32
    H.rth_thrender.argtypes = [vp, i32, C.POINTER(rt_camera), u32, u32, vp, C.c_uint64, i32, cp, i32, C.POINTER(rt_stats)]
    H.rth_read_image.restype = i32
    H.rth_read_image.argtypes = [cp, vp, C.c_uint64, C.POINTER(u32), C.POINTER(u32)]
    H.rth_write_png.restype = i32
    H.rth_write_png.argtypes = [cp, vp, u32, u32]
    _host = H
    return H


def last_error():
    return amd().rt_last_error().decode("utf-8", "replace")
