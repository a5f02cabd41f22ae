"""rust/raytracer-gpu/src/ffi.rs against include/rt_amd.h.  The image has no Rust toolchain, so the binding cannot be
compiled here; this test parses both files and fails when a struct (field names, order, types), a function (name,
argument order, argument and return types) or a constant differs.  It also checks that the reference-side file
implements the hook for every type of the reference that needs it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_amd.h")
CRATE = os.path.join(ROOT, "rust", "raytracer-gpu")

SCALARS = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "double": "f64", "int": "c_int",
           "char": "c_char", "void": "c_void"}


def rust_name(c_struct):            # rt_render_opts -> RtRenderOpts
    return "".join(p.capitalize() for p in c_struct.split("_"))


def c_type_to_rust(t):
    """'const rt_scene* const*' -> '*const *const RtScene', 'double*' -> '*mut f64', 'uint32_t' -> 'u32'."""
    t = t.strip()
    out = ""
    while t.endswith("*") or t.endswith("* const"):
        t = t[:-len("* const")].strip() if t.endswith("* const") else t[:-1].strip()
        # the pointee's constness: 'const X' or 'X const'
        if t.startswith("const ") and "*" not in t:
            out += "*const "
            t = t[len("const "):].strip()
        elif t.endswith(" const"):
            out += "*const "
            t = t[:-len(" const")].strip()
        else:
            out += "*mut "
    if t.startswith("const "):
        t = t[len("const "):].strip()
    base = SCALARS.get(t) or rust_name(t)
    return out + base


def parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    structs = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\} (\w+);", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, names = decl.split(None, 1)
            for nm in names.split(","):
                nm = nm.strip()
                m = re.match(r"(\w+)\[(\d+)\]$", nm)
                fields.append((m.group(1), f"[{SCALARS[ctype]}; {m.group(2)}]") if m else (nm, SCALARS[ctype]))
        structs[rust_name(name)] = fields
    funcs = {}
    flat = " ".join(src.split())
    protos = [re.search(r"((?:const )?\w+\*?) (rt_\w+)\(([^)]*)\)$", stmt.strip()) for stmt in flat.split(";")]
    for ret, name, args in [m.groups() for m in protos if m]:
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"(.*?)(\w+)$", a)
                params.append((m.group(2), c_type_to_rust(m.group(1))))
        funcs[name] = (params, None if ret == "void" else c_type_to_rust(ret))
    consts = {}
    for name, val in re.findall(r"#define (RT_\w+) \(?(-?\w+)\)?", src):
        if name == "RT_AMD_H":
            continue
        consts[name] = int(val.rstrip("u"), 0)
    for body in re.findall(r"enum \w+ \{(.*?)\};", src, flags=re.S):
        for name, val in re.findall(r"(RT_\w+) = (\d+)", body):
            consts[name] = int(val)
    return structs, funcs, consts


def parse_ffi():
    src = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    src = re.sub(r"//.*", "", src)
    structs = {}
    for name, body in re.findall(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", src, flags=re.S):
        structs[name] = [(n, t.strip()) for n, t in re.findall(r"pub (\w+): ([^,\n]+),", body)]
    funcs = {}
    block = re.search(r'extern "C" \{(.*)\n\}', src, flags=re.S).group(1)
    for name, args, ret in re.findall(r"pub fn (\w+)\(([^)]*)\)(?: -> ([^;]+))?;", block):
        params = [(p.split(":")[0].strip(), p.split(":", 1)[1].strip()) for p in args.split(",") if p.strip()]
        funcs[name] = (params, ret.strip() if ret else None)
    consts = {n: int(v.replace("_", ""), 0) for n, v in re.findall(r"pub const (RT_\w+): \w+ = (-?\w+);", src)}
    return structs, funcs, consts


def test_structs_match_the_header():
    h, _, _ = parse_header()
    r, _, _ = parse_ffi()
    assert set(h) == {"RtMaterial", "RtTexture", "RtXformOp", "RtChain", "RtMedium", "RtNode", "RtCamera", "RtRenderOpts", "RtStats"}
    for name, fields in h.items():
        assert r.get(name) == fields, f"{name}: header {fields} != ffi.rs {r.get(name)}"
    assert r["RtScene"] == []   # opaque


def test_functions_match_the_header():
    _, h, _ = parse_header()
    _, r, _ = parse_ffi()
    assert len(h) >= 24 and "rt_render" in h and "rt_intersect_batch_noalloc" in h
    assert set(h) == set(r), (set(h) ^ set(r))
    for name, (params, ret) in h.items():
        rp, rr = r[name]
        assert [t for _, t in rp] == [t for _, t in params], f"{name}: argument types {rp} != {params}"
        assert [n for n, _ in rp] == [n for n, _ in params], f"{name}: argument names {rp} != {params}"
        assert rr == ret, f"{name}: returns {rr}, header says {ret}"


def test_constants_match_the_header():
    _, _, h = parse_header()
    _, _, r = parse_ffi()
    assert len(h) > 25
    for name, val in h.items():
        assert name in r, f"{name} missing from ffi.rs"
        assert r[name] == val or r[name] == val & 0xFFFFFFFF, f"{name}: {r[name]} != {val}"


def test_every_reference_type_has_its_hook():
    src = open(os.path.join(CRATE, "reference", "gpu_impls.rs")).read()
    hittables = ["Sphere", "Parallelogram", "Triangle", "Quadric", "HittableSlice", "BVHNode", "Translate", "RotateY", "ConstantMedium"]
    materials = ["Lambertian", "Metal", "Dielectric", "DiffuseLight", "Isotropic"]
    textures = ["SolidColor", "CheckerTexture", "ImageTexture", "PerlinNoiseTexture"]
    for trait, types in (("FlattenHittable", hittables), ("FlattenMaterial", materials), ("FlattenTexture", textures)):
        for t in types:
            assert re.search(rf"impl {trait} for {t} \{{", src), f"no {trait} for {t}"
    assert "pub fn render_gpu(&self, world: &dyn Hittable)" in src
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    for mod in ("builder", "ffi", "flatten"):
        assert f"pub mod {mod};" in lib and os.path.exists(os.path.join(CRATE, "src", mod + ".rs"))
    # every rt_scene_set_* call of the header is made by SceneBuilder::commit
    builder = open(os.path.join(CRATE, "src", "builder.rs")).read()
    _, funcs, _ = parse_header()
    for name in funcs:
        if name.startswith("rt_scene_set_") or name in ("rt_scene_create", "rt_scene_commit", "rt_scene_destroy", "rt_render", "rt_tonemap_rgb8"):
            assert name + "(" in builder, f"builder.rs never calls {name}"
