"""The Rust shim (shim/vello_hip) is source only -- no Rust toolchain in the image -- so the one thing a compiler could
not check either is checked here: every item of its `extern "C"` block against include/vello_hip.h (function names,
arity, argument and return types, struct fields, constants).  Round 4 adds what can be checked of the rest without a
compiler: the vello_tests patch applies to the reference tree (`git apply --check`), the crate names no item of `vello`
that only exists under its "wgpu" feature (it is built with default-features = false), and `render_to_buffer` calls no
`&self` method while `Resolver::resolve`'s loans are alive."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vello_hip.h")
FFI = os.path.join(ROOT, "shim", "vello_hip", "src", "ffi.rs")

C_TO_RUST = {
    "int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "void": "()", "float": "f32", "uint8_t": "u8", "char": "c_char",
}


def strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def c_type_to_rust(t):
    """'const vello_hip_layout *' -> '*const vello_hip_layout'; 'void *const *' -> '*const *mut c_void'."""
    t = t.strip()
    t = re.sub(r"\[[^\]]*\]$", "*", t)          # array parameter = pointer
    toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\*", t)
    # split into base (up to the first *) and pointer levels with their trailing const
    base, levels, i = [], [], 0
    while i < len(toks) and toks[i] != "*":
        base.append(toks[i]); i += 1
    base_const = "const" in base
    base = [b for b in base if b != "const"]
    assert len(base) == 1, t
    name = base[0]
    rust = C_TO_RUST.get(name, name)
    if name == "void" and i < len(toks):
        rust = "c_void"
    pointee_const = base_const
    while i < len(toks):
        assert toks[i] == "*"; i += 1
        this_const = False
        if i < len(toks) and toks[i] == "const":
            this_const = True; i += 1
        rust = ("*const " if pointee_const else "*mut ") + rust
        pointee_const = this_const
    return rust


def parse_header():
    src = strip_c_comments(open(HEADER).read())
    funcs = {}
    for m in re.finditer(r"\n\s*((?:const\s+)?[A-Za-z_][A-Za-z_0-9]*(?:\s*\*)*)\s*(vello_hip_[a-z_0-9]+)\s*\(([^;{]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        arg_types = []
        if args.strip() and args.strip() != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)\s*(\[[^\]]*\])?$", a, flags=re.S)
                ty = am.group(1) + (am.group(3) or "")
                arg_types.append(c_type_to_rust(ty))
        funcs[name] = (c_type_to_rust(ret) if ret.strip() != "void" else "()", arg_types)
    structs = {}
    for m in re.finditer(r"typedef struct (vello_hip_[a-z_]+) \{(.*?)\} \1;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if decl:
                ty, names = decl.split(None, 1)
                fields += [(n.strip(), C_TO_RUST[ty]) for n in names.split(",")]
        structs[m.group(1)] = fields
    consts = {}
    for m in re.finditer(r"enum\s*\{(.*?)\};", src, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = [x.strip() for x in item.split("=")]
                nxt = int(v, 0)
            else:
                k = item
            consts[k] = nxt
            nxt += 1
    return funcs, structs, consts


def parse_rust():
    src = re.sub(r"//[^\n]*", "", open(FFI).read())
    funcs = {}
    for m in re.finditer(r"pub fn (vello_hip_[a-z_0-9]+)\((.*?)\)\s*(?:->\s*([^;]+))?;", src, flags=re.S):
        args = [a.split(":", 1)[1].strip() for a in m.group(2).split(",") if a.strip()]
        funcs[m.group(1)] = ((m.group(3) or "()").strip(), args)
    structs = {}
    for m in re.finditer(r"pub struct (vello_hip_[a-z_]+) \{(.*?)\}", src, flags=re.S):
        fields = [(f.split(":")[0].replace("pub", "").strip(), f.split(":")[1].strip()) for f in m.group(2).split(",") if ":" in f]
        structs[m.group(1)] = fields
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (VELLO_HIP_[A-Z_0-9]+): \w+ = (-?\d+);", src)}
    return funcs, structs, consts


def test_rust_ffi_matches_the_c_header():
    cf, cs, cc = parse_header()
    rf, rs, rc = parse_rust()
    assert len(cf) >= 30, sorted(cf)
    assert set(cf) == set(rf), (sorted(set(cf) - set(rf)), sorted(set(rf) - set(cf)))
    for name in sorted(cf):
        assert cf[name] == rf[name], f"{name}: header {cf[name]} vs shim {rf[name]}"
    for name, fields in cs.items():
        assert rs.get(name) == fields, f"struct {name}: header {fields} vs shim {rs.get(name)}"
    assert rs["vello_hip_ctx"] == [("_private", "[u8; 0]")]
    for k, v in rc.items():
        assert cc.get(k) == v, f"constant {k}: header {cc.get(k)} vs shim {v}"
    for k in ("VELLO_HIP_AA_AREA", "VELLO_HIP_AA_MSAA16", "VELLO_HIP_E_CAPACITY", "VELLO_HIP_DEBUG_NO_CULL", "VELLO_HIP_DEBUG_STROKE_KERNEL", "VELLO_HIP_DEBUG_SEQ_CLIP", "VELLO_HIP_DEBUG_FINE_SLICES", "VELLO_HIP_STAGE_COUNT"):
        assert k in rc


def test_shim_crate_tree_is_complete():
    for rel in ("Cargo.toml", "build.rs", "src/lib.rs", "src/ffi.rs"):
        assert os.path.exists(os.path.join(ROOT, "shim", "vello_hip", rel)), rel
    lib = open(os.path.join(ROOT, "shim", "vello_hip", "src", "lib.rs")).read()
    for item in ("pub struct HipRenderer", "pub fn new(", "pub fn render_to_buffer(", "pub fn render_to_vec(", "impl Drop for HipRenderer"):
        assert item in lib, item
    assert os.path.exists(os.path.join(ROOT, "shim", "vello_tests_patch", "render_then_debug.patch"))


def test_library_exports_every_function_of_the_shim(built):
    import vello_amd

    lib = vello_amd.load_library()
    rf, _, _ = parse_rust()
    for name in rf:
        assert hasattr(lib, name), name


REFERENCE = "/root/reference"
LIB_RS = os.path.join(ROOT, "shim", "vello_hip", "src", "lib.rs")
PATCH = os.path.join(ROOT, "shim", "vello_tests_patch", "render_then_debug.patch")


def test_vello_tests_patch_is_a_unified_diff():
    """Every hunk header carries ranges and the counts add up (what `git apply` checks first), reference or not."""
    lines = open(PATCH).read().split("\n")
    files = [l for l in lines if l.startswith("--- a/")]
    assert files == ["--- a/vello_tests/Cargo.toml", "--- a/vello_tests/src/lib.rs"], files
    i, hunks = 0, 0
    while i < len(lines):
        m = re.match(r"^@@ -(\d+),(\d+) \+(\d+),(\d+) @@", lines[i])
        if lines[i].startswith("@@"):
            assert m, f"hunk header without ranges: {lines[i]!r}"
            old, new = int(m.group(2)), int(m.group(4))
            i += 1
            while old or new:
                c = lines[i][:1]
                assert c in (" ", "-", "+"), f"line {i + 1}: {lines[i]!r}"
                old -= c in (" ", "-")
                new -= c in (" ", "+")
                i += 1
            assert old == 0 and new == 0
            hunks += 1
        else:
            i += 1
    assert hunks >= 4


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "vello_tests")) or shutil.which("git") is None,
                    reason="the reference tree is not on this machine")
def test_vello_tests_patch_applies_to_the_reference():
    r = subprocess.run(["git", "-C", REFERENCE, "apply", "--check", "--verbose", PATCH], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _rust_code(path):
    """The file without // comments and string literals' contents."""
    src = re.sub(r"//[^\n]*", "", open(path).read())
    return re.sub(r'"(?:[^"\\]|\\.)*"', '""', src)


def test_shim_uses_only_ungated_vello_items():
    code = _rust_code(LIB_RS)
    cargo = open(os.path.join(ROOT, "shim", "vello_hip", "Cargo.toml")).read()
    assert re.search(r'^vello = \{[^}]*default-features = false', cargo, flags=re.M)
    # items behind #[cfg(feature = "wgpu")] in vello/src/lib.rs (:112-115, :141-143, :371-430): not nameable in this build
    for gated in ("RendererOptions", "Renderer::", "vello::Renderer", "vello::util", "vello::wgpu", "wgpu::"):
        assert gated not in code, gated
    uses = re.search(r"use vello::\{([^}]*)\};", code).group(1)
    assert {u.strip() for u in uses.split(",")} == {"AaConfig", "AaSupport", "RenderParams", "Scene"}
    if os.path.isdir(os.path.join(REFERENCE, "vello", "src")):
        lib = open(os.path.join(REFERENCE, "vello", "src", "lib.rs")).read()
        for item in ("pub enum AaConfig", "pub struct AaSupport", "pub struct RenderParams", "pub use scene::{DrawGlyphs, Scene};"):
            at = lib.index(item)
            before = lib[:at].rstrip().split("\n")
            # the attribute lines directly above the item (doc comments skipped) must not gate it
            k = len(before) - 1
            while k >= 0 and before[k].lstrip().startswith(("///", "#[derive", "#[doc")):
                k -= 1
            assert "cfg(feature" not in before[k], (item, before[k])


def test_no_self_method_call_while_the_resolver_is_lent_out():
    """`Resolver::resolve<'a>(&'a mut self, ..) -> (Layout, Ramps<'a>, Images<'a>)` (vello_encoding/src/resolve.rs:183-187)
    keeps `self.resolver` mutably borrowed while `ramps` / `images` are used; a `self.method(..)` call in that region borrows
    all of `self` and is E0502.  Field accesses (`self.packed`, `self.atlas_size`) are disjoint borrows and fine."""
    code = _rust_code(LIB_RS)
    body = code[code.index("pub fn render_to_buffer"):code.index("pub fn render_to_vec")]
    after = body[body.index("self.resolver.resolve("):]
    last_use = max(after.rfind("ramps."), after.rfind("images."))
    region = after[len("self.resolver.resolve("):last_use]
    calls = re.findall(r"\bself\.([a-z_]+)\s*\(", region)
    assert calls == [], calls
    fields = set(re.findall(r"\bself\.([a-z_]+)\b", region))
    assert fields <= {"packed", "atlas_size"}, fields
    assert "fn error(ctx: *mut vello_hip_ctx" in code  # the error path takes the raw pointer, not &self
