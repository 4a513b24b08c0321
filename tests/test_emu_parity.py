"""Kernel-logic CI without a GPU: the UNMODIFIED kernel sources of vello_amd/csrc/engine, compiled by g++
against the SIMT emulator (tests/simt_emu), must agree with the CPU oracle stage by stage.  This checks
indexing / scan / allocation logic; memory-model behaviour is only checked on hardware (test_gpu_parity.py)."""
import os
import sys

import numpy as np
import pytest

import workloads
from tests.emu_lib import emu_library_path
from tests.parity import compare_frame
from vello_amd import AaConfig, Layout

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLACK, WHITE = 0xFF000000, 0xFFFFFFFF


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16])
def test_emu_smoke_circle(emu_engine, aa):
    packed, layout = workloads.smoke_circle_scene().resolve()
    compare_frame(emu_engine, packed, layout, 20, 20, BLACK, aa, f"emu_circle_{int(aa)}")


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa16])
def test_emu_stroke_styles(emu_engine, aa):
    packed, layout = workloads.stroke_styles_scene().resolve()
    compare_frame(emu_engine, packed, layout, 256, 256, WHITE, aa, f"emu_strokes_{int(aa)}")


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa8])
def test_emu_clip_blend(emu_engine, aa):
    packed, layout = workloads.clip_blend_scene().resolve()
    compare_frame(emu_engine, packed, layout, 256, 256, BLACK, aa, f"emu_clips_{int(aa)}")


@pytest.mark.parametrize("seed", [1, 2])
def test_emu_random_scene_multi_partition(emu_engine, seed):
    # > 256 draw objects and > 4096 tags: several look-back partitions, several coarse batches
    packed, layout = workloads.random_test_scene(seed, n_paths=700, size=384.0, strokes=True, clips=True).resolve()
    assert layout.n_draw_objects > 256 and (layout.path_data_base - layout.path_tag_base) > 1024
    compare_frame(emu_engine, packed, layout, 384, 384, BLACK, AaConfig.Msaa16, f"emu_random_{seed}")


def test_emu_tiger_small(emu_engine):
    d = np.load(os.path.join(GOLD, "tiger_scene.npz"))
    layout = Layout(*[int(v) for v in d["layout"]])
    # the fixture is encoded for a 1024 fit; render its top-left 320x320 window
    compare_frame(emu_engine, d["packed"], layout, 320, 320, WHITE, AaConfig.Msaa8, "emu_tiger")


def test_emu_capacity_overflow_leaves_target_untouched(built):
    # bump protocol: overflow sets bump.failed, fine does not touch the target, counters keep the demand
    import vello_amd
    import vello_amd._lib as L

    L._use_library(emu_library_path())
    try:
        eng = vello_amd.Engine(capacities={"lines": 64, "seg_counts": 64, "segments": 64})
        packed, layout = workloads.stroke_styles_scene().resolve()
        img, bump = eng.render(packed, layout, 256, 256, WHITE, AaConfig.Area)
        assert bump["failed"] != 0 and bump["lines"] > 64
        assert eng.sync() == -4
    finally:
        L._use_library(None)


@pytest.mark.parametrize("spare", [0, 1])
def test_emu_pools_exactly_full(built, spare):
    # every pool sized to what the frame takes (+ spare elements): the last tile, line, segment and PTCL word sit at the END of their
    # buffers -- coarse's two-word tile-bit windows, the 16-byte zero fill of an odd tile range, path_count's and path_tiling's gathers read
    # right up to it.  (The AddressSanitizer sweep, scripts/asan_check.sh, runs this suite: a read one element too far fails there.)
    import vello_amd
    import vello_amd._lib as L

    packed, layout = workloads.random_test_scene(9, n_paths=260, size=300.0, strokes=True, clips=True).resolve()
    L._use_library(emu_library_path())
    try:
        probe = vello_amd.Engine()
        _, need = probe.render(packed, layout, 300, 300, BLACK, AaConfig.Msaa16)
        assert need["failed"] == 0
        caps = {"lines": need["lines"] + spare, "tiles": need["tile"] + spare, "seg_counts": need["seg_counts"] + spare,
                "segments": need["segments"] + spare, "bin_data": need["binning"] + layout.bin_data_start + spare}
        eng = vello_amd.Engine(capacities=caps)
        got = eng.capacities()
        assert got["tiles"] == need["tile"] + spare and got["lines"] == need["lines"] + spare
        for aa in (AaConfig.Msaa16, AaConfig.Area):
            compare_frame(eng, packed, layout, 300, 300, BLACK, aa, f"emu_exact_pools_{spare}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0)
    finally:
        L._use_library(None)


def test_emu_frames_in_flight_rotate_lanes(built):
    # host logic of vello_hip_set_frames_in_flight: the ring of private buffer sets, sync_frame ages,
    # read_buffer following the newest lane
    import vello_amd
    import vello_amd._lib as L
    from oracle.oracle import Oracle

    L._use_library(emu_library_path())
    try:
        eng = vello_amd.Engine()
        packed, layout = workloads.stroke_styles_scene().resolve()
        eng.upload_scene(packed, layout)
        eng.set_frames_in_flight(3)
        o = Oracle()
        variants = [(96, 96, BLACK, AaConfig.Area), (128, 64, WHITE, AaConfig.Msaa8), (64, 128, BLACK, AaConfig.Msaa16)]
        for i in range(5):
            w, h, base, aa = variants[i % 3]
            eng.render_resident(w, h, base, aa)
            eng.sync_frame(0)
            img = eng.read_buffer("output", np.uint8, w * h * 4).reshape(h, w, 4)
            o.set_scene(packed, layout, w, h, base, int(aa))
            ref = o.render()
            assert np.abs(img.astype(int) - ref.astype(int)).max() <= (1 if aa == AaConfig.Area else 0)
        assert eng.sync() == 0
        with pytest.raises(Exception):
            eng.sync_frame(3)
        eng.set_frames_in_flight(1)                # shrink: back to one frame at a time
        with pytest.raises(Exception):
            eng.sync_frame(1)
        eng.render_resident(96, 96, BLACK, AaConfig.Area)
        assert eng.sync() == 0
    finally:
        L._use_library(None)


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa16])
def test_emu_gradient_image_blur_brushes(emu_engine, aa):
    # late-bound resources end to end: Resolver (ramps, atlas placement, draw-data patches) -> draw_leaf info ->
    # coarse commands -> fine's gradient / image / blurred-rounded-rect arms
    import vello_amd

    r = vello_amd.Resolver().resolve(workloads.brushes_scene())
    assert r.ramps is not None and r.atlas_size == 1024 and len(r.uploads) == 3
    compare_frame(emu_engine, r.packed, r.layout, 256, 256, WHITE, aa, f"emu_brushes_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                  resolved=r)


def test_emu_smoke_brush_goldens(emu_engine):
    # the kernel sources (g++ / SIMT emulator build) against the reference's gradient and image smoke snapshots
    import os

    import vello_amd
    from vello_amd import Extend

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "smoke_goldens.npz"))
    for pre in (True, False):
        r = vello_amd.Resolver().resolve(workloads.smoke_gradient_alpha_scene(pre))
        img, _, _ = compare_frame(emu_engine, r.packed, r.layout, 100, 50, WHITE, AaConfig.Area, f"emu_grad_alpha_{int(pre)}", tol=1,
                                  resolved=r)
        assert np.array_equal(img[:, :, :3], gold["gradient_color_alpha_" + ("premultiplied" if pre else "unpremultiplied")])
    rgba, rgb = gold["data_image_roundtrip_rgba"], gold["data_image_roundtrip_rgb"]
    for ext in (Extend.Pad, Extend.Reflect, Extend.Repeat):
        r = vello_amd.Resolver().resolve(workloads.smoke_data_image_scene(rgba, ext))
        img, _, _ = compare_frame(emu_engine, r.packed, r.layout, 31, 31, BLACK, AaConfig.Area, f"emu_data_image_{int(ext)}", tol=1,
                                  resolved=r)
        assert np.array_equal(img[:, :, :3], rgb)


def test_emu_property_images(emu_engine):
    # vello_tests/tests/property.rs:107-199 on the kernel sources: Bgra8 bytes and premultiplied Rgba8 bytes come out as the colours
    # they encode (the oracle is pinned to the same two tests in test_oracle_golden.py)
    import vello_amd
    from tests.test_oracle_golden import _premultiplied_distance

    for kind, base, eps in (("bgra", BLACK, 1e-4), ("premultiplied", 0x00000000, 1e-2)):
        scene, want = workloads.property_image_scene(kind)
        r = vello_amd.Resolver().resolve(scene)
        img, _, _ = compare_frame(emu_engine, r.packed, r.layout, 2, 2, base, AaConfig.Area, f"emu_property_{kind}", tol=1, resolved=r)
        assert (_premultiplied_distance(img, want) <= eps).all(), (kind, img.reshape(-1, 4))


def test_emu_auto_grow_reruns_until_the_frame_fits(built):
    # pools start far too small; robust mode grows lines -> seg_counts/segments -> ... round by round
    # (a failed stage hides the demand of the later ones) and the final frame equals the oracle's
    import vello_amd
    import vello_amd._lib as L
    from oracle.oracle import Oracle

    L._use_library(emu_library_path())
    try:
        tiny = {"lines": 64, "seg_counts": 64, "segments": 64, "tiles": 256, "bin_data": 512, "ptcl": 64 * 256 + 512, "blend_spill": 16}
        eng = vello_amd.Engine(capacities=tiny)
        packed, layout = workloads.clip_blend_scene().resolve()
        img, bump = eng.render(packed, layout, 256, 256, WHITE, AaConfig.Msaa8)
        assert bump["failed"] != 0 and eng.sync() == -4
        # manual protocol: get_bump -> grow_pools -> render again
        rounds = 0
        while bump["failed"] != 0:
            assert eng.grow_pools(bump), bump
            img, bump = eng.render(packed, layout, 256, 256, WHITE, AaConfig.Msaa8)
            rounds += 1
            assert rounds < 8
        assert rounds >= 2
        o = Oracle()
        o.set_scene(packed, layout, 256, 256, WHITE, int(AaConfig.Msaa8))
        assert np.array_equal(img, o.render())
        caps = eng.capacities()
        assert caps["lines"] >= bump["lines"] and caps["segments"] >= caps["seg_counts"] >= bump["seg_counts"]
        assert not eng.grow_pools(bump)          # nothing left to grow
        # automatic: one blocking call
        eng2 = vello_amd.Engine(capacities=tiny)
        eng2.set_auto_grow(True)
        img2, bump2 = eng2.render(packed, layout, 256, 256, WHITE, AaConfig.Msaa8)
        assert bump2["failed"] == 0 and np.array_equal(img2, img)
    finally:
        L._use_library(None)


def test_unorm8_conversion_without_division_is_exact(built):
    # fine.hip's unorm8_to_f32 (q = b*r; e = fma(-q, 255, b); q + e*r) must equal (float)b / 255.0f for every byte
    f32 = np.float32
    r = f32(1.0) / f32(255.0)
    for b in range(256):
        x = f32(b)
        q = f32(x * r)
        e = f32(np.float64(x) - np.float64(q) * 255.0)          # exact in fp64: 24-bit q times 8-bit constant
        got = f32(np.float64(q) + np.float64(e) * np.float64(r))  # |e*r| << ulp(q)/2 margin: no double-rounding hazard here
        assert got == f32(x / f32(255.0)), b


def test_emu_flatten_staging_overflow(emu_engine):
    # one flatten workgroup producing more lines than its LDS staging area (3072): the overflow pieces are written
    # straight to the soup; counts, multiset of lines and the image must not change
    packed, layout = workloads.heavy_strokes_scene().resolve()
    img, ref, bump = compare_frame(emu_engine, packed, layout, 1024, 1024, BLACK, AaConfig.Msaa8, "emu_heavy_strokes")
    assert bump["lines"] > 3072


@pytest.mark.parametrize("which", ["tricky_strokes", "fill_types", "robust_paths", "funky_paths"])
def test_emu_reference_test_scenes(emu_engine, which):
    # scenes of the reference's own catalogue (examples/scenes/src/test_scenes.rs:513-770, :1610-1691)
    scene, w, h = getattr(workloads, which + "_scene")()
    packed, layout = scene.resolve()
    compare_frame(emu_engine, packed, layout, w, h, BLACK, AaConfig.Msaa16, "emu_" + which)


def test_emu_render_frame_keeps_a_scene_per_lane(built):
    # vello_hip_render_frame: every in-flight frame has its OWN scene (animation form); lanes must not see each other's
    import vello_amd
    import vello_amd._lib as L
    from oracle.oracle import Oracle

    L._use_library(emu_library_path())
    try:
        eng = vello_amd.Engine()
        eng.set_frames_in_flight(3)
        scenes = [workloads.stroke_styles_scene(), workloads.clip_blend_scene(), workloads.smoke_circle_scene(),
                  workloads.random_test_scene(3, n_paths=60, size=128.0)]
        o = Oracle()
        for rep in range(2):
            for k, sc in enumerate(scenes):
                packed, layout = sc.resolve()
                w, h, aa = (128, 128, AaConfig.Msaa8) if k != 2 else (20, 20, AaConfig.Msaa16)
                eng.render_frame(packed, layout, w, h, BLACK, aa)
                eng.sync_frame(0)
                img = eng.read_buffer("output", np.uint8, w * h * 4).reshape(h, w, 4)
                o.set_scene(packed, layout, w, h, BLACK, int(aa))
                assert np.array_equal(img, o.render()), (rep, k)
        assert eng.sync() == 0
        # the shared-scene form takes over again
        packed, layout = scenes[0].resolve()
        eng.upload_scene(packed, layout)
        eng.render_resident(128, 128, WHITE, AaConfig.Msaa8)
        eng.sync_frame(0)
        o.set_scene(packed, layout, 128, 128, WHITE, int(AaConfig.Msaa8))
        assert np.array_equal(eng.read_buffer("output", np.uint8, 128 * 128 * 4).reshape(128, 128, 4), o.render())
    finally:
        L._use_library(None)


@pytest.mark.parametrize("which", ["gradient_extend", "blend_grid", "deep_blend", "many_clips", "blurred_rounded_rect", "image_sampling",
                                   "image_sampling_bicubic"])
def test_emu_reference_brush_and_layer_scenes(emu_engine, which):
    # test_scenes.rs:978-1043, :1213-1304, :1398-1436: gradient extend modes, the 16 mix modes over gradients, blend
    # stack spill, 600 clip layers
    import vello_amd

    scene, w, h = getattr(workloads, which + "_scene")()
    r = vello_amd.Resolver().resolve(scene)
    compare_frame(emu_engine, r.packed, r.layout, w, h, BLACK, AaConfig.Msaa8, "emu_" + which, resolved=r)


@pytest.mark.parametrize("which", ["ref_stroke_styles", "ref_stroke_styles_non_uniform", "ref_stroke_styles_skew", "two_point_radial",
                                   "conflation_artifacts", "labyrinth", "clip_test", "luminance_mask", "image_extend_modes",
                                   "image_extend_modes_nearest", "brush_transform"])
def test_emu_reference_catalogue_second_batch(emu_engine, which):
    # test_scenes.rs:335-511 (cap / join / miter-limit matrix under identity, non-uniform scale and skew), :1045-1211
    # (COLR two-point radial cases x extend modes), :1444-1531 (shared edges of opposite winding), :1533-1608 (140
    # overlapping sub-paths in one fill), :1708-1911 (even-odd clip, STROKE-styled clip layer, blend layer cut by its clip
    # rect), :2214-2289 (luminance mask layer), :2168-2212 (image brush extend modes)
    import vello_amd
    from vello_amd import Affine, ImageQuality

    if which.startswith("ref_stroke_styles"):
        t = {"ref_stroke_styles": None, "ref_stroke_styles_non_uniform": Affine.scale_non_uniform(1.2, 0.7),
             "ref_stroke_styles_skew": Affine.skew(1.0, 0.0)}[which]
        scene, w, h = workloads.ref_stroke_styles_scene(t)
    elif which == "image_extend_modes_nearest":
        scene, w, h = workloads.image_extend_modes_scene(ImageQuality.Low)
    else:
        scene, w, h = getattr(workloads, which + "_scene")()
    r = vello_amd.Resolver().resolve(scene)
    for aa in (AaConfig.Area, AaConfig.Msaa16):
        compare_frame(emu_engine, r.packed, r.layout, w, h, WHITE, aa, f"emu_{which}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                      resolved=r)


def test_emu_reference_catalogue_third_batch(emu_engine):
    # test_scenes.rs:2291-2349 (an IMAGE under a luminance-mask layer; the JPEG asset replaced by a synthetic image of the same role,
    # the whole scene at a quarter of its size for the emulator) and :1693-1706 (a translucent square over the base colour, which the
    # reference animates through the hues: four base colours, one of them translucent)
    import vello_amd

    scene, w, h = workloads.image_luminance_mask_scene(scale=4)
    r = vello_amd.Resolver().resolve(scene)
    for aa in (AaConfig.Area, AaConfig.Msaa16):
        compare_frame(emu_engine, r.packed, r.layout, w, h, WHITE, aa, f"emu_image_luminance_mask_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                      resolved=r)
    scene, w, h = workloads.base_color_test_scene()
    r = vello_amd.Resolver().resolve(scene)
    for base in (0xFF3C8CE6, 0xFF20C040, 0xFFFFFFFF, 0x80402010):
        compare_frame(emu_engine, r.packed, r.layout, 160, 160, base, AaConfig.Msaa8, f"emu_base_color_{base:08x}", resolved=r)


def test_emu_image_atlas_residency_sequence(emu_engine):
    # resolve.rs:507-541 / image_cache.rs end to end: a 32-texel atlas that has to evict, repack and grow while frames keep
    # sampling it; after every resolve the frame must equal the oracle's rendering of the same atlas state
    import vello_amd
    from vello_amd import Affine, ImageBrush, ImageData, ImageQuality, Scene

    def img(w, h, seed):
        rng = np.random.default_rng(seed)
        px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        px[:, :, 3] = 255
        return ImageData(px)

    def scene_of(images):
        s = Scene()
        for k, im in enumerate(images):
            s.draw_image(ImageBrush(im, quality=ImageQuality.Low), Affine.translate(4.0 + 34.0 * k, 4.0) * Affine.scale(2.0))
        return s

    res = vello_amd.Resolver(atlas_sizes=(32, 64))
    a, b, c, d = img(16, 16, 1), img(16, 16, 2), img(16, 16, 3), img(32, 16, 4)
    frames = [[a, b], [a, b, c], [c], [c], [c], [c, d, a], [a, b, c, d, img(32, 32, 5)]]
    seen_evict = seen_grow = False
    for i, images in enumerate(frames):
        r = res.resolve(scene_of(images))
        seen_evict |= r.evicted > 0
        seen_grow |= r.atlas_size > 32
        compare_frame(emu_engine, r.packed, r.layout, 200, 80, WHITE, AaConfig.Area, f"emu_residency_{i}", tol=1, resolved=r)
    assert seen_evict and seen_grow


def test_emu_fuzz_whole_api(emu_engine):
    # workloads/fuzz.py: seeded random scenes over the whole scene API (every brush kind with degenerate parameters, all
    # mix / compose modes, luminance masks, fill- and stroke-styled clips, zero / hairline / huge stroke widths, skewed,
    # mirrored and near-singular affines, repeated points, geometry far off screen and on tile corners)
    import vello_amd
    from workloads.fuzz import fuzz_scene

    emu_engine.set_auto_grow(True)
    try:
        for seed in range(0, 60):
            r = vello_amd.Resolver().resolve(fuzz_scene(seed))
            aa = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16][seed % 3]
            base = [0xFF000000, 0xFFFFFFFF, 0x00000000, 0x80FF8040][seed % 4]
            compare_frame(emu_engine, r.packed, r.layout, 128, 128, base, aa, f"emu_fuzz_{seed}", tol=1 if aa == AaConfig.Area else 0, resolved=r,
                          order_sensitive=True)
    finally:
        emu_engine.set_auto_grow(False)


def test_emu_fuzz_target_sizes_and_long_scenes(emu_engine):
    # the same generator at awkward target sizes (one pixel, one tile, not multiples of 16, several bins) and with up to
    # 700 operations per scene (several 256-draw batches per bin in coarse)
    import vello_amd
    from workloads.fuzz import fuzz_scene

    emu_engine.set_auto_grow(True)
    try:
        for seed in range(0, 36):
            w, h = [(333, 205), (97, 530), (512, 512), (16, 16), (1, 1), (700, 40)][seed % 6]
            r = vello_amd.Resolver().resolve(fuzz_scene(seed, size=max(w, h, 8), n_ops=[40, 700, 300][seed % 3]))
            aa = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16][(seed // 2) % 3]
            base = [0xFF000000, 0xFFFFFFFF, 0x00000000, 0x80FF8040][seed % 4]
            compare_frame(emu_engine, r.packed, r.layout, w, h, base, aa, f"emu_fuzz2_{seed}", tol=1 if aa == AaConfig.Area else 0, resolved=r,
                          order_sensitive=True)
    finally:
        emu_engine.set_auto_grow(False)


def test_emu_fuzz_auto_grow_from_tiny_pools(built):
    # robust dynamic memory under fuzz: every pool starts at 64 elements, auto-grow has to find the frame's
    # demand stage by stage (a failed stage hides the demand of the later ones) and the final frame must equal the oracle's
    import vello_amd
    import vello_amd._lib as L
    from workloads.fuzz import fuzz_scene

    L._use_library(emu_library_path())
    try:
        for seed in range(0, 30):
            eng = vello_amd.Engine(capacities={"lines": 64, "binning": 64, "tile": 64, "seg_counts": 64, "segments": 64, "blend": 16,
                                               "ptcl": 64 * 4 + 64})
            eng.set_auto_grow(True)
            w, h = [(128, 128), (300, 200), (64, 64)][seed % 3]
            r = vello_amd.Resolver().resolve(fuzz_scene(seed, size=max(w, h), n_ops=[40, 300][seed % 2]))
            aa = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16][(seed // 2) % 3]
            compare_frame(eng, r.packed, r.layout, w, h, 0xFF203040, aa, f"emu_fuzz3_{seed}", tol=1 if aa == AaConfig.Area else 0, resolved=r,
                          order_sensitive=True)
            del eng
    finally:
        L._use_library(None)


def test_emu_inconsistent_scenes_are_refused(emu_engine):
    # A packed scene whose streams contradict each other (WebGPU's robust buffer access absorbs this upstream; HIP has
    # none, so the engine has to refuse it -- never read or write outside its buffers -- and stay usable afterwards)
    import vello_amd

    good_packed, layout = workloads.stroke_styles_scene().resolve()
    n_tag_bytes = (layout.path_data_base - layout.path_tag_base) * 4

    def corrupt_tags(byte):
        p = good_packed.copy()
        p[layout.path_tag_base * 4: layout.path_tag_base * 4 + n_tag_bytes] = byte
        return p

    def corrupt_draw_tags(word):
        p = good_packed.copy()
        p.view(np.uint32)[layout.draw_tag_base: layout.draw_tag_base + layout.n_draw_objects] = word
        return p

    cases = {
        "every tag a f32 cubic: more path data than the stream holds": (corrupt_tags(0x0B), layout),
        "every tag a TRANSFORM marker": (corrupt_tags(0x20), layout),
        "every tag a STYLE marker": (corrupt_tags(0x40), layout),
        "every draw object a radial gradient: more draw data / info than the layout holds": (corrupt_draw_tags(0x29C), layout),
        "clip tags without clips in the layout": (corrupt_draw_tags(0x49), layout),
        "more draw objects than paths": (good_packed, layout._replace(n_draw_objects=layout.n_paths + 5)),
        # fewer clip tags than n_clips: k_clip would walk clip_inp entries draw_leaf never wrote
        "more clips in the layout than clip tags": (good_packed, layout._replace(n_clips=layout.n_clips + 3)),
    }
    for what, (packed, lay) in cases.items():
        with pytest.raises(vello_amd.VelloHipError):
            emu_engine.render(packed, lay, 256, 256, WHITE, AaConfig.Msaa8)
        # the context is not poisoned: the next (valid) frame is right
        compare_frame(emu_engine, good_packed, layout, 256, 256, WHITE, AaConfig.Msaa8, "emu_invalid_after_" + what.split(":")[0].replace(" ", "_"))


def test_emu_thousands_of_segments_in_one_tile(emu_engine):
    # fills and hairline strokes of 70 ... 2000 random segments crammed into a couple of tiles: the > 64-segment and
    # > 512-crossing paths of fine's MSAA (segment batches, record overflow) and counters far beyond one byte per sample
    import math

    from vello_amd import Affine, BezPath, Color, Fill, Scene, Stroke

    rng = np.random.default_rng(7)
    emu_engine.set_auto_grow(True)
    try:
        for case in range(6):
            s = Scene()
            for _ in range(int(rng.integers(1, 6))):
                p = BezPath()
                cx, cy = rng.uniform(4, 28, 2)
                p.move_to((cx, cy))
                for _ in range(int(rng.choice([70, 130, 600, 2000]))):
                    a, r = rng.uniform(0, 2 * math.pi), rng.uniform(0.2, 9.0)
                    end = (cx + r * math.cos(a), cy + r * math.sin(a))
                    if rng.random() < 0.7:
                        p.line_to(end)
                    else:
                        b = rng.uniform(0, 2 * math.pi)
                        p.quad_to((cx + r * math.cos(b), cy + r * math.sin(b)), end)
                p.close_path()
                col = Color(float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.choice([1.0, 0.5])))
                if rng.random() < 0.3:
                    s.stroke(Stroke(float(rng.uniform(0.05, 0.6))), Affine.IDENTITY, col, None, p)
                else:
                    s.fill(Fill(int(rng.integers(0, 2))), Affine.IDENTITY, col, None, p)
            packed, layout = s.resolve()
            for aa in (AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16):
                compare_frame(emu_engine, packed, layout, 48, 48, 0xFF101010, aa, f"emu_dense_{case}_{int(aa)}",
                              tol=1 if aa == AaConfig.Area else 0, order_sensitive=True)
    finally:
        emu_engine.set_auto_grow(False)


@pytest.mark.parametrize("kind", ["clip", "blend"])
def test_emu_layers_nested_300_deep(emu_engine, kind):
    # clip_leaf's stack beyond its LDS window (spill), fine's blend stack far beyond BLEND_STACK_SPLIT, the blend spill pool
    # grown past the reference's 2^20 entries by auto-grow
    from oracle.oracle import Oracle
    from vello_amd import Affine, BlendMode, Circle, Color, Compose, Fill, Mix, Rect, Scene

    depth = 300
    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(30, 60, 200), None, Rect(0, 0, 96, 96))
    for d in range(depth):
        shape = Rect(0.01 * d, 0.02 * d, 96 - 0.01 * d, 96 - 0.015 * d) if d % 3 else Circle((48, 48), 60 - 0.01 * d)
        if kind == "clip":
            s.push_clip_layer(Fill.NonZero, Affine.IDENTITY, shape)
        else:
            s.push_layer(Fill.NonZero, BlendMode(Mix(d % 16), Compose.SrcOver), 0.97, Affine.IDENTITY, shape)
        if d % 50 == 0:
            s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgba8(255, 200, (d * 7) % 255, 120), None,
                   Circle((20 + d % 60, 30 + (d // 7) % 40), 14))
    s.fill(Fill.EvenOdd, Affine.IDENTITY, Color.from_rgb8(250, 250, 20), None, Circle((48, 48), 30))
    for _ in range(depth):
        s.pop_layer()
    packed, layout = s.resolve()
    emu_engine.set_auto_grow(True)
    try:
        for aa in (AaConfig.Area, AaConfig.Msaa16):
            _, _, bump = compare_frame(emu_engine, packed, layout, 96, 96, BLACK, aa, f"emu_deep_{kind}_{int(aa)}",
                                       tol=1 if aa == AaConfig.Area else 0, order_sensitive=True, oracle=Oracle(capacity_scale=16))
        assert bump["blend"] > (1 << 20)
    finally:
        emu_engine.set_auto_grow(False)


@pytest.mark.parametrize("v", [1e6, 1e9, 3e38, float("inf"), float("nan"), -1e9, -3e38])
def test_emu_extreme_coordinates(emu_engine, v):
    # Coordinates, stroke widths and transforms from a million pixels off screen to f32's limits, infinities and NaN.  What
    # the reference computes there is deterministic IEEE arithmetic and the kernels must reproduce it; what it LEAVES to
    # WebGPU's robust buffer access must not become a wild access here: a line that starts more than 65 535 tile
    # crossings outside the viewport overflows the 16-bit crossing index of SegmentCount (path_count.wgsl:196) and
    # path_tiling recomputes a tile that is not the path's (found with these inputs: the oracle crashed, the engine read
    # out of bounds).  Also found here: the exact straight-segment shortcut has to stand back when |h|^2 overflows in the
    # general loop (chords > 3e9 px), and atan2(inf, inf) is pi/4, not inf / inf.
    from oracle.oracle import Oracle
    from vello_amd import Affine, BezPath, Color, Fill, Rect, Scene, Stroke

    def scene(kind):
        s = Scene()
        s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(10, 200, 30), None, Rect(8, 8, 56, 56))
        p = BezPath()
        if kind == 0:
            p.move_to((10.0, 10.0)); p.line_to((v, 20.0)); p.line_to((30.0, v)); p.close_path()
            s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(200, 0, 0), None, p)
        elif kind == 1:
            p.move_to((10.0, 10.0)); p.curve_to((v, 5.0), (20.0, v), (40.0, 40.0))
            s.stroke(Stroke(3.0), Affine.IDENTITY, Color.from_rgb8(200, 0, 200), None, p)
        elif kind == 2:
            p.move_to((10.0, 10.0)); p.quad_to((30.0, 50.0), (50.0, 10.0))
            s.stroke(Stroke(abs(v) if v == v else v), Affine.IDENTITY, Color.from_rgb8(0, 0, 200), None, p)
        else:
            p.move_to((10.0, 10.0)); p.line_to((50.0, 12.0)); p.line_to((30.0, 50.0)); p.close_path()
            s.fill(Fill.EvenOdd, Affine.scale(v) if kind == 3 else Affine.translate(v, 0.0), Color.from_rgb8(0, 200, 200), None, p)
        return s

    emu_engine.set_auto_grow(True)
    try:
        for kind in range(5):
            packed, layout = scene(kind).resolve()
            for aa in (AaConfig.Area, AaConfig.Msaa16):
                compare_frame(emu_engine, packed, layout, 64, 64, BLACK, aa, f"emu_extreme_{kind}_{v}_{int(aa)}",
                              tol=1 if aa == AaConfig.Area else 0, order_sensitive=True, min_agree=None, back_half=False, oracle=Oracle(capacity_scale=4, auto_grow=True))
    finally:
        emu_engine.set_auto_grow(False)


def test_emu_fuzz_extreme_values(emu_engine):
    # the whole-API fuzzer with 12 % of its points drawn from {+-1e6 ... +-3e38, +-inf, NaN}
    import vello_amd
    from oracle.oracle import Oracle
    from workloads.fuzz import fuzz_scene

    emu_engine.set_auto_grow(True)
    try:
        for seed in [s for s in range(0, 40) if s not in (2, 5, 25)]:  # those three emit 6-7 million lines each (15-30 s)
            r = vello_amd.Resolver().resolve(fuzz_scene(seed, n_ops=14, extreme=True))
            aa = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16][seed % 3]
            compare_frame(emu_engine, r.packed, r.layout, 128, 128, BLACK, aa, f"emu_fuzzx_{seed}", tol=1 if aa == AaConfig.Area else 0,
                          resolved=r, order_sensitive=True, min_agree=None, back_half=False, oracle=Oracle(capacity_scale=4, auto_grow=True))
    finally:
        emu_engine.set_auto_grow(False)


def stroke_kernel_untame_inputs(eng, name, seeds=(4552, 8707, 11797)):
    """The fuzz seeds whose paths of NaN lines got another box from the FORCED stroke kernel than from the oracle (a NaN makes min / max
    depend on the order of their operands, and the kernel writes a tag's lines in another order than flatten.wgsl's one invocation): since
    round 6 the kernel hands inputs that are not finite, or large enough to overflow, to the heavy code.  Every stage against the oracle,
    with the kernel forced, one frame at a time and with frames in flight (the kernel as a launch of its own)."""
    import vello_amd
    from oracle.oracle import Oracle
    from workloads.fuzz import fuzz_scene

    eng.set_auto_grow(True)
    try:
        for seed in seeds:
            r = vello_amd.Resolver().resolve(fuzz_scene(seed, n_ops=14, extreme=True))
            aa = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16][seed % 3]
            for nif in (1, 2):
                eng.set_frames_in_flight(nif)
                eng.update_debug_flags(stroke_kernel=True)
                compare_frame(eng, r.packed, r.layout, 128, 128, BLACK, aa, f"{name}_{seed}_{nif}", tol=1 if aa == AaConfig.Area else 0,
                              resolved=r, order_sensitive=True, min_agree=None, back_half=False, oracle=Oracle(capacity_scale=4, auto_grow=True))
    finally:
        eng.update_debug_flags(stroke_kernel=False)
        eng.set_frames_in_flight(1)
        eng.set_auto_grow(False)


def test_emu_stroke_kernel_untame_inputs(emu_engine):
    stroke_kernel_untame_inputs(emu_engine, "emu_stroke_untame")


@pytest.mark.parametrize("atlas", [None, (32, 128)])
def test_emu_persistent_resolver_over_many_frames(emu_engine, atlas):
    # ONE Resolver across 100 frames of recurring fuzz scenes (60 distinct ones): ramp ids reused and evicted in the ramp cache
    # (ramp_cache.rs:26-63, at most 64 retained), images resident, dirty, evicted, repacked and the atlas grown in the image
    # cache -- every frame against the oracle's rendering of the same resolved state
    import vello_amd
    from workloads.fuzz import fuzz_scene

    res = vello_amd.Resolver() if atlas is None else vello_amd.Resolver(atlas_sizes=atlas)
    evicted = 0
    emu_engine.set_auto_grow(True)
    try:
        for i in range(100):
            seed = 5000 + (i * 7) % 60
            r = res.resolve(fuzz_scene(seed))
            evicted += r.evicted
            aa = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16][seed % 3]
            compare_frame(emu_engine, r.packed, r.layout, 128, 128, 0xFF102030, aa, f"emu_seq_{i}", tol=1 if aa == AaConfig.Area else 0,
                          resolved=r, order_sensitive=True)
    finally:
        emu_engine.set_auto_grow(False)
    assert (evicted > 0) == (atlas is not None)


def test_emu_reference_regression_scenes(emu_engine):
    # the reference's own regression tests that need no fonts: known_issues.rs:54-90 (clip_blends, issue #1198),
    # regression.rs:18-31 (rounded_rectangle_watertight, issue #616), :107-121 (stroke_width_zero, issue #662), known_issues.rs:20-52
    # (layer_size, issue #1061, `should_panic` upstream: an EMPTY Compose::Clear layer over a red square -- held to the restated shaders' frame)
    import vello_amd

    cases = [workloads.clip_blends_scene() + ("clip_blends",)] + workloads.regression_stroke_scenes()
    for scene, w, h, name in cases:
        r = vello_amd.Resolver().resolve(scene)
        for aa in (AaConfig.Area, AaConfig.Msaa16):
            img, _, _ = compare_frame(emu_engine, r.packed, r.layout, w, h, BLACK, aa, f"emu_{name}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                                      resolved=r)
        if name == "stroke_width_zero":
            assert (img[:, :, :3] == 0).all()
        if name == "clip_blends":
            assert tuple(img[5, 5]) == (0, 0, 255, 255) and tuple(img[90, 50]) == (0, 0, 212, 255)   # blue x aquamarine, multiplied
        if name == "layer_size":  # (the Clear layer's rectangle comes out transparent -- over black: 0 -- the green around it stays)
            assert tuple(img[5, 5]) == (0, 255, 0, 255) and tuple(img[30, 30]) == (0, 0, 0, 0)


def test_emu_zero_width_stroke_clip_before_any_transform(emu_engine):
    # scene.rs:179-183 as the FIRST operation of a scene: the zero-width stroke clip encodes a style and an empty path
    # but no transform, so its tags carry trans_ix = 0 - 1.  WGSL indexes in u32 (the read lands just below
    # transform_base); pointer arithmetic in two steps lands 24 GB away (found by the fuzzer, seed 80).
    import vello_amd
    from vello_amd import Affine, Circle, Color, Fill, Rect, Scene, Stroke

    s = Scene()
    s.push_clip_layer(Stroke(0.0), Affine.IDENTITY, Circle((40.0, 40.0), 20.0))
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(0, 255, 0), None, Rect(0.0, 0.0, 80.0, 80.0))
    s.pop_layer()
    s.fill(Fill.NonZero, Affine.translate(3.0, 4.0), Color.from_rgb8(255, 0, 0), None, Rect(10.0, 10.0, 30.0, 30.0))
    r = vello_amd.Resolver().resolve(s)
    img, _, _ = compare_frame(emu_engine, r.packed, r.layout, 80, 80, BLACK, AaConfig.Msaa8, "emu_zero_width_clip_first", resolved=r)
    assert (img[:, :, 1] == 0).all(), "the zero-width stroke clip suppresses everything inside the layer"
    assert tuple(img[20, 20]) == (255, 0, 0, 255)


def test_emu_auto_grow_covers_large_targets(built):
    # the PTCL pool holds a fixed 64 words per tile: a target with more tiles than the pool was sized for is an
    # E_INVALID configuration error, unless robust mode may grow the pool
    import vello_amd
    import vello_amd._lib as L
    from oracle.oracle import Oracle

    L._use_library(emu_library_path())
    try:
        eng = vello_amd.Engine(capacities={"ptcl": 64 * 16 + 512})          # enough for 16 tiles only
        packed, layout = workloads.stroke_styles_scene().resolve()
        with pytest.raises(vello_amd.VelloHipError):
            eng.render(packed, layout, 256, 256, WHITE, AaConfig.Msaa8)       # 256 tiles
        eng.set_auto_grow(True)
        img, bump = eng.render(packed, layout, 256, 256, WHITE, AaConfig.Msaa8)
        assert bump["failed"] == 0 and eng.capacities()["ptcl"] >= 64 * 256
        o = Oracle()
        o.set_scene(packed, layout, 256, 256, WHITE, int(AaConfig.Msaa8))
        assert np.array_equal(img, o.render())
    finally:
        L._use_library(None)


def test_emu_auto_grow_with_frames_in_flight(built):
    # an overflow on lane A followed by the auto-grow retry on lane B must report the RETRY's outcome; a lane
    # dropped from the rotation must not keep failing vello_hip_sync
    import vello_amd
    import vello_amd._lib as L
    from oracle.oracle import Oracle

    L._use_library(emu_library_path())
    try:
        eng = vello_amd.Engine(capacities={"lines": 64, "seg_counts": 64, "segments": 64, "tiles": 64})
        eng.set_frames_in_flight(3)
        eng.set_auto_grow(True)
        packed, layout = workloads.stroke_styles_scene().resolve()
        o = Oracle()
        o.set_scene(packed, layout, 256, 256, WHITE, int(AaConfig.Msaa8))
        ref = o.render()
        for _ in range(2):
            img, bump = eng.render(packed, layout, 256, 256, WHITE, AaConfig.Msaa8)
            assert bump["failed"] == 0 and np.array_equal(img, ref)
            assert eng.sync() == 0
        # without auto-grow: overflow, then shrink the rotation so that the failed lane drops out of it
        eng2 = vello_amd.Engine(capacities={"lines": 64, "seg_counts": 64, "segments": 64})
        eng2.set_frames_in_flight(3)
        eng2.upload_scene(packed, layout)
        for _ in range(3):
            eng2.render_resident(256, 256, WHITE, AaConfig.Msaa8)
        assert eng2.sync() == -4
        eng2.set_frames_in_flight(1)
        small, small_layout = workloads.smoke_square_scene().resolve()
        eng2.upload_scene(small, small_layout)
        eng2.render_resident(20, 20, BLACK, AaConfig.Area)
        assert eng2.sync() == 0, "lanes outside the rotation must not fail the frames of the rotation"
    finally:
        L._use_library(None)


def test_emu_resident_frames_show_the_uploaded_scene(built):
    # after vello_hip_render_frame (private per-lane scenes) a resident frame renders the scene of
    # vello_hip_upload_scene on whichever lane the rotation reaches -- or fails if there is none
    import vello_amd
    import vello_amd._lib as L
    from oracle.oracle import Oracle

    L._use_library(emu_library_path())
    try:
        eng = vello_amd.Engine()
        eng.set_frames_in_flight(2)
        a, la = workloads.stroke_styles_scene().resolve()
        b, lb = workloads.clip_blend_scene().resolve()
        eng.render_frame(b, lb, 128, 128, BLACK, AaConfig.Msaa8)
        with pytest.raises(vello_amd.VelloHipError):
            eng.render_resident(128, 128, BLACK, AaConfig.Msaa8)   # nothing was ever uploaded as THE scene
        eng.sync()
        eng.upload_scene(a, la)
        eng.render_frame(b, lb, 128, 128, BLACK, AaConfig.Msaa8)
        eng.render_frame(b, lb, 128, 128, BLACK, AaConfig.Msaa8)   # both lanes now hold private scene b
        o = Oracle()
        o.set_scene(a, la, 128, 128, BLACK, int(AaConfig.Msaa8))
        ref = o.render()
        for _ in range(3):
            eng.render_resident(128, 128, BLACK, AaConfig.Msaa8)
            eng.sync_frame(0)
            assert np.array_equal(eng.read_buffer("output", np.uint8, 128 * 128 * 4).reshape(128, 128, 4), ref)
        assert eng.sync() == 0
    finally:
        L._use_library(None)


def test_emu_target_buffers_are_validated(emu_engine):
    # a target smaller than the frame must be refused before the C ABI is called
    packed, layout = workloads.smoke_square_scene().resolve()
    emu_engine.upload_scene(packed, layout)
    import torch

    small = torch.zeros((10, 20, 4), dtype=torch.uint8)
    with pytest.raises((ValueError, AssertionError)):
        emu_engine.render_resident(20, 20, BLACK, AaConfig.Area, out=small)
    with pytest.raises((ValueError, AssertionError)):
        emu_engine.render_resident(20, 20, BLACK, AaConfig.Area, out=torch.zeros((20, 20, 4), dtype=torch.float32))


def test_emu_estimator_presizes_the_pools(built):
    # SURVEY 8f f4, second half (vello_encoding/src/estimate.rs): the estimate covers the real demand of every pool, and
    # with it robust mode renders from 64-element pools in ONE round instead of one round per overflowing stage
    import vello_amd
    import vello_amd._lib as L
    from oracle.oracle import Oracle
    from vello_amd.renderer import estimate_capacities

    L._use_library(emu_library_path())
    try:
        tiger = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiger_scene.npz"))
        cases = [workloads.stroke_styles_scene().resolve() + (256, 256), workloads.clip_blend_scene().resolve() + (256, 256),
                 workloads.random_test_scene(11, n_paths=150, size=300.0, strokes=True, clips=True).resolve() + (300, 300),
                 workloads.heavy_strokes_scene().resolve() + (1024, 1024),
                 (tiger["packed"], vello_amd.Layout(*[int(v) for v in tiger["layout"]]), 1024, 1024)]
        for packed, layout, w, h in cases:
            o = Oracle(capacity_scale=2)
            o.set_scene(packed, layout, w, h, WHITE, int(AaConfig.Msaa8))
            ref = o.render()
            need = o.bump()
            est = estimate_capacities(packed, layout, w, h)
            assert est["lines"] >= need["lines"] and est["seg_counts"] >= need["seg_counts"] and est["segments"] >= need["segments"]
            assert est["tiles"] >= need["tile"] and est["bin_data"] >= layout.bin_data_start + need["binning"]
            tiny = {"lines": 64, "seg_counts": 64, "segments": 64, "tiles": 64, "bin_data": layout.bin_data_start + 64,
                    "ptcl": 64 * ((w + 15) // 16) * ((h + 15) // 16) + 512}
            eng = vello_amd.Engine(capacities=tiny)
            eng.set_auto_grow(True)
            img, bump = eng.render(packed, layout, w, h, WHITE, AaConfig.Msaa8)
            assert bump["failed"] == 0 and np.array_equal(img, ref)
            assert eng.last_render_attempts() == 1, (eng.last_render_attempts(), est, need)
    finally:
        L._use_library(None)


def test_emu_gather_frames_between_contexts(built):
    # vello_hip_gather_frames (single-process form): two contexts, each frame lands in its slot of the
    # destination; here both "GPUs" are the emulated device, on hardware the same call runs peer copies over xGMI
    import vello_amd
    import vello_amd._lib as L
    from oracle.oracle import Oracle
    from vello_amd.renderer import gather_frames

    L._use_library(emu_library_path())
    try:
        scenes = [workloads.stroke_styles_scene(), workloads.clip_blend_scene()]
        engines, srcs, refs = [], [], []
        for sc in scenes:
            packed, layout = sc.resolve()
            e = vello_amd.Engine()
            e.upload_scene(packed, layout)
            src = np.zeros((128, 128, 4), dtype=np.uint8)
            e._check(e._lib.vello_hip_render_resident(e._h, __import__("ctypes").byref(e._params(128, 128, BLACK, AaConfig.Msaa8)),
                                                      src.ctypes.data, 128 * 4), "render_resident")
            o = Oracle()
            o.set_scene(packed, layout, 128, 128, BLACK, int(AaConfig.Msaa8))
            refs.append(o.render())
            engines.append(e)
            srcs.append(src)
        dst = np.zeros((2, 128, 128, 4), dtype=np.uint8)
        gather_frames(engines, [s.ctypes.data for s in srcs], [dst[i].ctypes.data for i in range(2)], 128 * 128 * 4)
        for i in range(2):
            assert np.array_equal(dst[i], refs[i])
    finally:
        L._use_library(None)


def _fill_rule_interleave_scene(seed):
    """One or two tiles covered by many small fills whose rule alternates at random between NonZero and EvenOdd, with
    a > 64-segment fill (taken one at a time by fine) and empty-coverage fills thrown in: every order in which fine's
    three MSAA resolve paths (sparse non-zero, dense even-odd, unbatched) can hand the sample counters to each other."""
    import math

    from vello_amd import Affine, BezPath, Color, Fill, Scene

    rng = np.random.default_rng(seed)
    s = Scene()
    for k in range(40):
        p = BezPath()
        big = rng.random() < 0.12
        n = int(rng.integers(200, 320)) if big else int(rng.integers(3, 9))
        cx, cy = rng.uniform(2, 30, 2)
        r0 = float(rng.uniform(2.0, 6.0)) if big else float(rng.uniform(1.0, 9.0))
        for i in range(n):
            a = 2 * math.pi * (i * (3 if rng.random() < 0.5 else 1)) / n  # star polygons self-intersect: the rules differ
            r = r0 * (1.0 if i % 2 == 0 else float(rng.uniform(0.3, 1.0)))
            pt = (cx + r * math.cos(a), cy + r * math.sin(a))
            p.move_to(pt) if i == 0 else p.line_to(pt)
        p.close_path()
        if rng.random() < 0.1:  # a fill that misses every sample (degenerate sliver on a pixel boundary)
            p = BezPath()
            p.move_to((4.0, 8.0)); p.line_to((20.0, 8.0)); p.line_to((4.0, 8.0)); p.close_path()
        col = Color(float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.choice([1.0, 0.6])))
        s.fill(Fill(int(rng.integers(0, 2))), Affine.IDENTITY, col, None, p)
    return s.resolve()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_emu_fill_rules_interleaved_in_one_tile(emu_engine, seed):
    packed, layout = _fill_rule_interleave_scene(seed)
    for aa in (AaConfig.Msaa8, AaConfig.Msaa16):
        compare_frame(emu_engine, packed, layout, 32, 32, 0xFF203040, aa, f"emu_rules_{seed}_{int(aa)}", order_sensitive=True)


def _stroke_kernel_cases():
    """(name, packed, layout, w, h) of the stroke-heavy scenes: flatten's stroked-line kernel normally takes over from
    393 216 stroked lines, VELLO_HIP_DEBUG_STROKE_KERNEL runs it for these."""
    import math

    from vello_amd import Affine, BezPath, Cap, Color, Join, Scene, Stroke

    out = []
    s = workloads.stroke_styles_scene()
    out.append(("stroke_styles", *s.resolve(), 256, 256))
    for which in ("tricky_strokes", "robust_paths"):
        scene, w, h = getattr(workloads, which + "_scene")()
        out.append((which, *scene.resolve(), w, h))
    for name, t in (("ref_identity", None), ("ref_skew", Affine.skew(1.0, 0.0)), ("ref_non_uniform", Affine.scale_non_uniform(1.2, 0.7))):
        scene, w, h = workloads.ref_stroke_styles_scene(t)
        out.append((name, *scene.resolve(), w, h))
    # polylines with every join / cap, closed and open, zero-length and sub-ULP segments (those are handed on to the heavy kernel),
    # hairlines and a width far larger than the segments
    rng = np.random.default_rng(5)
    s = Scene()
    for k in range(60):
        p = BezPath()
        x, y = rng.uniform(8, 120, 2)
        p.move_to((x, y))
        for i in range(int(rng.integers(1, 14))):
            r = rng.random()
            if r < 0.08:
                pass  # zero-length segment
            elif r < 0.14:
                x, y = x + float(rng.choice([1e-7, -3e-6, 2e-5])), y
            else:
                a = rng.uniform(0, 2 * math.pi)
                d = float(rng.choice([0.3, 2.0, 9.0, 30.0]))
                x, y = x + d * math.cos(a), y + d * math.sin(a)
            p.line_to((x, y))
        if rng.random() < 0.4:
            p.close_path()
        st = Stroke(float(rng.choice([0.0, 0.05, 1.0, 3.5, 24.0])))
        st = st.with_join([Join.Bevel, Join.Miter, Join.Round][int(rng.integers(0, 3))]).with_caps([Cap.Butt, Cap.Square, Cap.Round][int(rng.integers(0, 3))])
        tr = Affine.IDENTITY if rng.random() < 0.6 else Affine.rotate(float(rng.uniform(0, 6))) * Affine.scale(float(rng.uniform(0.3, 2.5)))
        s.stroke(st, Affine.translate(64.0, 64.0) * tr * Affine.translate(-64.0, -64.0), Color(float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), 0.5, 0.8), None, p)
    out.append(("random_polylines", *s.resolve(), 128, 128))
    # nothing but sub-micro-pixel and zero-length stroked segments: every one of them is handed on to the heavy kernel's list
    s = Scene()
    for k in range(40):
        p = BezPath()
        x, y = rng.uniform(8, 56, 2)
        p.move_to((x, y))
        for i in range(12):
            x += float(rng.choice([0.0, 1e-7, -1e-7]))
            p.line_to((x, y))
        s.stroke(Stroke(float(rng.choice([1.0, 6.0]))).with_caps(Cap.Round), Affine.IDENTITY, Color(0.2, 0.5, float(rng.uniform(0, 1)), 1.0), None, p)
    out.append(("all_handed_on", *s.resolve(), 64, 64))
    return out


def path_count_both_forms(eng, name):
    """path_count cuts the line soup into chunks of 1 024 lines while the size of the scene's soup is unknown, and from then on
    into chunks of 256 when the soup is small (engine.h PATH_COUNT_SMALL_MAX_LINES): the first frame of a small scene runs one
    form, the frames after a finished one the other -- all against the oracle."""
    from oracle.oracle import Oracle

    packed, layout = workloads.mmark_scene(n=1500).resolve()
    eng.upload_scene(packed, layout)
    o = Oracle()
    o.set_scene(packed, layout, 1024, 1024, WHITE, int(AaConfig.Msaa16))
    ref = o.render()
    ob = o.bump()
    for k in range(3):
        eng.render_resident(1024, 1024, WHITE, AaConfig.Msaa16)
        eng.sync_frame(0)
        img = eng.read_buffer("output", np.uint8, 1024 * 1024 * 4).reshape(1024, 1024, 4)
        bump = eng.bump()
        assert np.array_equal(img, ref), f"{name}: frame {k} differs"
        assert bump["failed"] == 0 and all(bump[key] == ob[key] for key in ("tile", "seg_counts", "lines", "binning")), (name, k, bump, ob)
    # round 6: with frames in flight a soup of unknown or large size takes the form with the small LDS footprint (path.hip
    # PcInFlight: chunks of 512 lines, a table of 256 cache lines, a stash of 384) -- the first frame after an upload
    eng.set_frames_in_flight(2)
    try:
        eng.upload_scene(packed, layout)
        eng.render_resident(1024, 1024, WHITE, AaConfig.Msaa16)
        eng.sync_frame(0)
        img = eng.read_buffer("output", np.uint8, 1024 * 1024 * 4).reshape(1024, 1024, 4)
        bump = eng.bump()
        assert np.array_equal(img, ref), f"{name}: the frame of the in-flight form differs"
        assert bump["failed"] == 0 and all(bump[key] == ob[key] for key in ("tile", "seg_counts", "lines", "binning")), (name, bump, ob)
    finally:
        eng.set_frames_in_flight(1)


def test_emu_path_count_both_forms(emu_engine):
    path_count_both_forms(emu_engine, "emu_path_count_forms")


def path_count_long_lines(eng, name):
    """k_path_count's seldom-walked arms with the product's own constants: long lines give a wave far more crossings than its stash
    holds (768: the rest go straight to memory after the flush) and a chunk far more cache lines of tiles than its table has
    entries (512: those crossings take the returning global add in the counting pass); fills and strokes, both fill rules, a line
    that leaves the target on every side.  Everything against the oracle, the back half included."""
    from vello_amd import Affine, BezPath, Color, Fill, Scene, Stroke

    rng = np.random.default_rng(77)
    s = Scene()
    for k in range(40):
        p = BezPath()
        p.move_to((float(rng.uniform(-200, 1800)), float(rng.uniform(-200, 1800))))
        for _ in range(int(rng.integers(3, 9))):
            p.line_to((float(rng.uniform(-200, 1800)), float(rng.uniform(-200, 1800))))
        col = Color(float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0.3, 1)))
        if k % 3 == 0:
            s.stroke(Stroke(float(rng.uniform(1, 9))), Affine.IDENTITY, col, None, p)
        else:
            p.close_path()
            s.fill(Fill.EvenOdd if k % 2 else Fill.NonZero, Affine.IDENTITY, col, None, p)
    packed, layout = s.resolve()
    for aa in (AaConfig.Area, AaConfig.Msaa16):
        compare_frame(eng, packed, layout, 1600, 1600, BLACK, aa, f"{name}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0)
    # the same arms of the form frames in flight take (PcInFlight: a stash of 384, a table of 256 entries)
    eng.set_frames_in_flight(2)
    try:
        compare_frame(eng, packed, layout, 1600, 1600, BLACK, AaConfig.Msaa16, f"{name}_in_flight")
    finally:
        eng.set_frames_in_flight(1)


def test_emu_path_count_long_lines(emu_engine):
    path_count_long_lines(emu_engine, "emu_pc_long")


@pytest.mark.parametrize("case", range(8))
def test_emu_stroked_line_kernel(emu_engine, case):
    name, packed, layout, w, h = _stroke_kernel_cases()[case]
    emu_engine.set_debug_flags(stroke_kernel=True)
    emu_engine.set_auto_grow(True)
    try:
        # (one frame in flight under area AA -- every stage and the line soup against the oracle --, two under MSAA16; the GPU suite's
        # test_gpu_stroked_line_kernel renders both modes with one)
        compare_frame(emu_engine, packed, layout, w, h, WHITE, AaConfig.Area, f"emu_strokekernel_{name}_0", tol=1)
        # with frames in flight the stroke workgroups are a kernel of their own ahead of the heavy list's (k_flatten_strokes,
        # k_flatten_heavy) instead of part of its launch (k_flatten_main, k_flatten_tail)
        emu_engine.set_frames_in_flight(2)
        compare_frame(emu_engine, packed, layout, w, h, WHITE, AaConfig.Msaa16, f"emu_strokekernel_{name}_2inflight")
    finally:
        emu_engine.set_frames_in_flight(1)
        emu_engine.set_debug_flags()
        emu_engine.set_auto_grow(False)


def flatten_kernel_sets(engine, name, packed, layout, w, h, bg=WHITE, in_flight=True, aas=(AaConfig.Area, AaConfig.Msaa16)):
    """Both sets of kernels for flatten's heavy list -- the wave-cooperative walk (k_flatten_main<true> / k_flatten_heavy<true>)
    and every lane on its own (<false>, round 4's) -- forced on the same scene, one frame in flight (the stroke workgroups beside
    the heavy list's: k_flatten_main) and two (k_flatten_heavy): stages, line soup as a multiset and images against the oracle."""
    try:
        for which in ("flatten_coop", "flatten_alone"):
            engine.set_debug_flags(**{which: True})
            for aa in aas:
                compare_frame(engine, packed, layout, w, h, bg, aa, f"{name}_{which}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0)
            if in_flight:
                engine.set_frames_in_flight(2)
                compare_frame(engine, packed, layout, w, h, bg, AaConfig.Msaa16, f"{name}_{which}_2inflight")
                engine.set_frames_in_flight(1)
    finally:
        engine.set_frames_in_flight(1)
        engine.set_debug_flags()


@pytest.mark.parametrize("case", range(8))
def test_emu_flatten_kernel_sets(emu_engine, case):
    name, packed, layout, w, h = _stroke_kernel_cases()[case]
    emu_engine.set_auto_grow(True)
    try:
        # (the kernel sets differ in flatten alone: one frame in flight under area AA -- every stage and the line soup --, two under MSAA16;
        # the GPU suite renders both modes with one)
        flatten_kernel_sets(emu_engine, "emu_flsets_" + name, packed, layout, w, h, aas=(AaConfig.Area,))
    finally:
        emu_engine.set_auto_grow(False)


def test_emu_flatten_kernel_sets_curves(emu_engine):
    # fills' curves (the cardioid, funky paths) and stroked curves with every join / cap
    for name, fn in (("cardioid", workloads.cardioid_scene), ("funky", workloads.funky_paths_scene)):
        r = fn()
        s, w, h = r if isinstance(r, tuple) else (r, 512, 512)
        packed, layout = s.resolve()
        flatten_kernel_sets(emu_engine, "emu_flsets_" + name, packed, layout, w, h, in_flight=False, aas=(AaConfig.Area,))


@pytest.mark.parametrize("case", range(7))
def test_emu_front_fusion(emu_engine, case):
    # small scenes: the workgroups of consecutive stages as turns of one launch (k_front) -- here of ONE workgroup, the emulator's
    # launches run their workgroups one after the other; the grid barrier itself is the GPU suite's (test_gpu_front_fusion)
    from tests.parity import check_front_fusion, front_fusion_cases

    # (two frames in flight are host logic here -- the lanes' private buffers and k_front's barrier counter: the small cases carry it,
    # the two large ones, random_700 and the tiger, render with one)
    check_front_fusion(emu_engine, front_fusion_cases()[case], in_flight=(1, 2) if case < 5 else (1,))


def test_emu_clip_stage_partitioned(emu_engine):
    # a5: clip_reduce / clip_leaf as partitioned kernels (clip.hip) and as the one-wave stack machine, against the oracle's stack
    from tests.parity import clip_structures, compare_clip_stage

    for name, ops in clip_structures(big=False):
        compare_clip_stage(emu_engine, ops, np.random.default_rng(len(ops)), "emu clips " + name)
    compare_clip_stage(emu_engine, [1] * 3000 + [1, -1] * 32000 + [-1] * 3000, np.random.default_rng(3), "emu clips 70 000")


def test_emu_write_image_between_frames(emu_engine):
    # host logic of the stream-ordered vello_hip_write_image (staging blocks, epochs): every frame shows the upload before it
    import vello_amd
    from vello_amd import Affine, ImageBrush, ImageData, ImageQuality, Scene

    frames = [np.full((16, 16, 4), 255, dtype=np.uint8) for _ in range(6)]
    for k, px in enumerate(frames):
        px[:, :, 0] = 40 * k
        px[k, :, 1] = 3
    s = Scene()
    s.draw_image(ImageBrush(ImageData(frames[0]), quality=ImageQuality.Low), Affine.translate(4.0, 4.0) * Affine.scale(2.0))
    r = vello_amd.Resolver().resolve(s)
    (x, y, _), = r.uploads
    emu_engine.set_frames_in_flight(3)
    try:
        emu_engine.upload_resolved(r)
        for k in range(6):
            emu_engine.write_image(x, y, frames[k])
            emu_engine.render_resident(48, 48, BLACK, AaConfig.Area)
            emu_engine.sync_frame(0)
            img = emu_engine.read_buffer("output", np.uint8, 48 * 48 * 4).reshape(48, 48, 4)
            for ty in (0, k, 15):
                assert tuple(img[4 + 2 * ty + 1, 4 + 2 * 7 + 1]) == tuple(frames[k][ty, 7]), (k, ty)
    finally:
        emu_engine.set_frames_in_flight(1)


def _pipeline_cases():
    m = sys.modules[__name__]
    cases = [("tiger", lambda e: m.test_emu_tiger_small(e)),
             ("stroke_styles", lambda e: m.test_emu_stroke_styles(e, AaConfig.Msaa16)),
             ("clip_blend", lambda e: m.test_emu_clip_blend(e, AaConfig.Msaa8)),
             ("brushes", lambda e: m.test_emu_gradient_image_blur_brushes(e, AaConfig.Msaa16)),
             ("thousands_of_segments", lambda e: m.test_emu_thousands_of_segments_in_one_tile(e)),
             ("fuzz", lambda e: m.test_emu_fuzz_whole_api(e)),
             ("nested_clips", lambda e: m.test_emu_layers_nested_300_deep(e, "clip"))]
    cases += [(f"rules_{seed}", lambda e, seed=seed: m.test_emu_fill_rules_interleaved_in_one_tile(e, seed)) for seed in (11,)]
    cases += [(w, lambda e, w=w: m.test_emu_reference_brush_and_layer_scenes(e, w)) for w in ("blend_grid", "deep_blend", "many_clips")]
    cases += [(w, lambda e, w=w: m.test_emu_reference_test_scenes(e, w)) for w in ("fill_types",)]
    return cases


@pytest.mark.parametrize("name,body", _pipeline_cases(), ids=[c[0] for c in _pipeline_cases()])
def test_emu_fine_slices(emu_engine, name, body):
    # fine's sliced path (VELLO_HIP_DEBUG_FINE_SLICES: every tile's list cut into slices of 4 fills, coverage by one wave per
    # slice, the last one to finish composites) through the same oracle comparisons as the one-wave path: batches, fills that
    # do not fit a batch, both fill rules, clip / blend stacks beyond the register window, brushes, the fuzzers
    emu_engine.set_debug_flags(fine_slices=True)
    try:
        body(emu_engine)
        if name in ("tiger", "rules_11", "blend_grid"):  # (the flag did take these frames through the sliced path)
            assert emu_engine.fine_slice_stats()[0] > 0, "no tile was cut into slices"
    finally:
        emu_engine.set_debug_flags()


def test_emu_fine_slices_survive_split_stage_ranges(emu_engine):
    # vello_hip_run_stages seam: COARSE in one call, FINE in a later one.  Coarse cuts tiles into slices according to the
    # number of slice blocks fine is going to launch; between the two calls the engine learns the scene's demand (sync reads
    # the control block) and would size fine's grid differently -- fine must launch with what coarse was told.
    packed, layout = workloads.random_test_scene(4, n_paths=500, size=256.0, strokes=True, clips=True).resolve()
    emu_engine.set_debug_flags(fine_slices=True)
    try:
        ref, bump = emu_engine.render(packed, layout, 256, 256, BLACK, AaConfig.Msaa16)
        assert bump["failed"] == 0 and emu_engine.fine_slice_stats()[0] > 0
        emu_engine.upload_scene(packed, layout)   # (the demand is unknown again: a quarter of the tiles' worth of blocks)
        emu_engine.run_stages(256, 256, BLACK, AaConfig.Msaa16, "pathtag_scan", "coarse")
        assert emu_engine.sync() == 0             # (reads the control block: the demand is known now)
        emu_engine.run_stages(256, 256, BLACK, AaConfig.Msaa16, "path_tiling", "fine")
        assert emu_engine.sync() == 0
        img = emu_engine.read_buffer("output", np.uint8, 256 * 256 * 4).reshape(256, 256, 4)
        assert np.array_equal(img, ref)
    finally:
        emu_engine.set_debug_flags()
