"""Parity tests proper: the HIP engine on a real MI355X, called through the C ABI, vs the CPU oracle.

Bars: integer stages bit-exact (tag/draw monoids, bboxes, bump counters, tile backdrops,
MSAA coverage -> MSAA images bit-exact); the line soup equal as a multiset (order follows atomics);
area-AA images within 1 LSB per channel (f32 summation order over segments differs).
"""
import os

import numpy as np
import pytest

import workloads
from tests.parity import compare_frame
from vello_amd import AaConfig, Layout

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLACK, WHITE = 0xFF000000, 0xFFFFFFFF


def test_native_library_is_loaded(gpu_engine):
    import vello_amd

    maps = open("/proc/self/maps").read()
    assert "libvello_hip.so" in maps and vello_amd.library_path() in maps


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16])
def test_smoke_goldens_on_gpu(gpu_engine, aa):
    gold = np.load(os.path.join(GOLD, "smoke_goldens.npz"))
    packed, layout = workloads.smoke_circle_scene().resolve()
    img, _, bump = compare_frame(gpu_engine, packed, layout, 20, 20, BLACK, aa, f"gpu_circle_{int(aa)}", tol=0 if aa else 1)
    if aa == AaConfig.Area:
        assert np.abs(img[:, :, :3].astype(int) - gold["filled_circle"].astype(int)).max() <= 1
    packed, layout = workloads.smoke_square_scene().resolve()
    img, _, _ = compare_frame(gpu_engine, packed, layout, 20, 20, BLACK, aa, f"gpu_square_{int(aa)}", tol=0 if aa else 1)
    assert np.array_equal(img[:, :, :3], gold["filled_square"])


def test_smoke_brush_goldens_on_gpu(gpu_engine):
    # the reference's gradient / image smoke snapshots (regression.rs:33-104, :150-209): GPU == oracle == snapshot
    import vello_amd
    from vello_amd import Extend

    gold = np.load(os.path.join(GOLD, "smoke_goldens.npz"))
    for pre in (True, False):
        r = vello_amd.Resolver().resolve(workloads.smoke_gradient_alpha_scene(pre))
        img, _, _ = compare_frame(gpu_engine, r.packed, r.layout, 100, 50, WHITE, AaConfig.Area, f"gpu_grad_alpha_{int(pre)}", tol=1,
                                  resolved=r)
        name = "gradient_color_alpha_" + ("premultiplied" if pre else "unpremultiplied")
        assert np.array_equal(img[:, :, :3], gold[name]), name
    rgba, rgb = gold["data_image_roundtrip_rgba"], gold["data_image_roundtrip_rgb"]
    for ext in (Extend.Pad, Extend.Reflect, Extend.Repeat):
        r = vello_amd.Resolver().resolve(workloads.smoke_data_image_scene(rgba, ext))
        img, _, _ = compare_frame(gpu_engine, r.packed, r.layout, 31, 31, BLACK, AaConfig.Area, f"gpu_data_image_{int(ext)}", tol=1,
                                  resolved=r)
        assert np.array_equal(img[:, :, :3], rgb), f"data_image_roundtrip extend {ext}"


def test_config_c1_circle_256_area(gpu_engine):
    packed, layout = workloads.circle_scene().resolve()
    compare_frame(gpu_engine, packed, layout, 256, 256, BLACK, AaConfig.Area, "gpu_c1", tol=1)


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16])
def test_stroke_styles(gpu_engine, aa):
    packed, layout = workloads.stroke_styles_scene().resolve()
    compare_frame(gpu_engine, packed, layout, 256, 256, WHITE, aa, f"gpu_strokes_{int(aa)}", tol=0 if aa else 1)


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16])
def test_clip_blend_layers(gpu_engine, aa):
    packed, layout = workloads.clip_blend_scene().resolve()
    compare_frame(gpu_engine, packed, layout, 256, 256, BLACK, aa, f"gpu_clips_{int(aa)}", tol=0 if aa else 1)


def test_deep_blend_spill(gpu_engine):
    from vello_amd import Affine, BlendMode, Circle, Color, Compose, Fill, Mix, Rect, Scene

    s = Scene()
    for d in range(9):  # deeper than BLEND_STACK_SPLIT = 4 -> blend_spill path
        s.push_layer(Fill.NonZero, BlendMode(Mix.Multiply if d % 2 else Mix.Screen, Compose.SrcOver), 0.9, Affine.IDENTITY,
                     Circle((64.0 + d, 64.0), 60.0 - 3 * d))
        s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgba8(30 * d, 255 - 20 * d, 100, 200), None, Rect(10 + 4 * d, 10, 120 - 4 * d, 120))
    for d in range(9):
        s.pop_layer()
    packed, layout = s.resolve()
    _, _, bump = compare_frame(gpu_engine, packed, layout, 128, 128, BLACK, AaConfig.Msaa16, "gpu_deep_blend")
    assert bump["blend"] > 0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_scenes_multi_partition(gpu_engine, seed):
    packed, layout = workloads.random_test_scene(seed, n_paths=1500, size=768.0, strokes=True, clips=True).resolve()
    compare_frame(gpu_engine, packed, layout, 768, 768, BLACK, AaConfig.Msaa16, f"gpu_random_{seed}")
    compare_frame(gpu_engine, packed, layout, 768, 768, BLACK, AaConfig.Area, f"gpu_random_area_{seed}", tol=1, check_stages=False)


def test_config_c2_tiger_1024_msaa8(gpu_engine):
    d = np.load(os.path.join(GOLD, "tiger_scene.npz"))
    layout = Layout(*[int(v) for v in d["layout"]])
    compare_frame(gpu_engine, d["packed"], layout, 1024, 1024, WHITE, AaConfig.Msaa8, "gpu_tiger")


def test_many_bins_all_red(gpu_engine):
    # vello_tests/tests/regression.rs:213-254: > 256 bins
    from vello_amd import Affine, Color, Fill, Rect, Scene

    s = Scene()
    s.fill(Fill.NonZero, Affine.IDENTITY, Color.from_rgb8(255, 0, 0), None, Rect(0.0, 0.0, 4352.0, 4352.0))
    packed, layout = s.resolve()
    img, bump = gpu_engine.render(packed, layout, 4352, 4352, BLACK, AaConfig.Area)
    assert bump["failed"] == 0
    assert (img == np.array([255, 0, 0, 255], dtype=np.uint8)).all()


def test_empty_scene_is_base_color(gpu_engine):
    from vello_amd import Scene

    packed, layout = Scene().resolve()
    img, bump = gpu_engine.render(packed, layout, 33, 17, 0xFF336699, AaConfig.Msaa16)
    assert (img == np.array([0x99, 0x66, 0x33, 0xFF], dtype=np.uint8)).all()


def test_capacity_overflow_protocol(built):
    import vello_amd

    eng = vello_amd.Engine(capacities={"lines": 64, "seg_counts": 64, "segments": 64})
    packed, layout = workloads.stroke_styles_scene().resolve()
    img, bump = eng.render(packed, layout, 256, 256, WHITE, AaConfig.Area)
    assert bump["failed"] != 0 and bump["lines"] > 64 and eng.sync() == -4


def test_aa_mode_must_be_enabled(built):
    import vello_amd

    eng = vello_amd.Engine(aa_mask=1)  # area only
    packed, layout = workloads.smoke_circle_scene().resolve()
    with pytest.raises(vello_amd.VelloHipError, match="AA mode"):
        eng.render(packed, layout, 20, 20, BLACK, AaConfig.Msaa16)


def test_renderer_api_render_to_texture(built):
    # the reference-shaped API: Renderer::render_to_texture(scene, texture, params), host and device targets
    import torch

    import vello_amd
    from oracle.oracle import Oracle
    from vello_amd import Color, Renderer, RenderParams

    scene = workloads.stroke_styles_scene()
    r = Renderer()
    params = RenderParams(Color.from_rgb8(255, 255, 255), 256, 256, AaConfig.Msaa16)
    host = np.zeros((256, 256, 4), dtype=np.uint8)
    r.render_to_texture(scene, host, params)
    dev = torch.zeros((256, 256, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()  # (torch's zero fill runs on torch's stream, the engine on its own)
    r.render_to_texture(scene, dev, params)
    torch.cuda.synchronize()
    packed, layout = scene.resolve()
    o = Oracle()
    o.set_scene(packed, layout, 256, 256, WHITE, 2)
    ref = o.render()
    assert np.array_equal(host, ref) and np.array_equal(dev.cpu().numpy(), ref)


def test_repeated_frames_are_deterministic_images(gpu_engine):
    # look-back state / bump reset every frame; MSAA output must not depend on atomic order
    packed, layout = workloads.random_test_scene(11, n_paths=800, size=512.0).resolve()
    first, _ = gpu_engine.render(packed, layout, 512, 512, BLACK, AaConfig.Msaa16)
    for _ in range(5):
        again, bump = gpu_engine.render(packed, layout, 512, 512, BLACK, AaConfig.Msaa16)
        assert bump["failed"] == 0 and np.array_equal(first, again)


def test_config_c3_paris_like_full_size(gpu_engine):
    # BASELINE config C3 at full size against the oracle (a few seconds of CPU) + size-independent properties
    from oracle.oracle import Oracle

    scene = workloads.paris_like_scene()
    packed, layout = scene.resolve()
    o = Oracle()
    img, ref, bump = compare_frame(gpu_engine, packed, layout, 1600, 1600, WHITE, AaConfig.Msaa16, "gpu_paris", oracle=o)
    assert bump["failed"] == 0
    assert (img[:, :, 3] == 255).all()  # opaque base + opaque paints


def test_config_c3_d2_scene_full_size(built):
    # BASELINE config C3 as SURVEY 8d d2 restates it (70 % stroked polylines / 25 % polygons / 5 % blobs, steps
    # 4-40 px): 3.06 M lines, 4.28 M crossings -- beyond the reference's fixed pools, so the context is created with
    # bench.py's D2_CAPS.  Every intermediate incl. bin lists, SegmentCounts, PTCL and segment slices vs the oracle.
    import bench
    import vello_amd
    from oracle.oracle import Oracle

    packed, layout = workloads.paris_like_scene_d2().resolve()
    eng = vello_amd.Engine(device=0, capacities=bench.D2_CAPS)
    img, ref, bump = compare_frame(eng, packed, layout, 1600, 1600, WHITE, AaConfig.Msaa16, "gpu_paris_d2", oracle=Oracle(capacity_scale=8))
    assert bump["failed"] == 0 and bump["lines"] > (1 << 21) and bump["seg_counts"] > (1 << 21)
    assert (img[:, :, 3] == 255).all()
    # the scene has tiles of >= 64 fills: fine cut them into slices (engine.h FINE_SLICE_FILLS) -- the last frame compare_frame
    # rendered is the NO_CULL one, whose lists are the longest
    items, cov_words = eng.fine_slice_stats()
    assert items > 100 and cov_words > 0, "no tile of the d2 scene went through fine's sliced path"
    # with frames in flight flatten launches the stroked lines' workgroups as a kernel of their own (k_flatten_strokes ->
    # k_flatten_heavy) instead of beside the heavy list's (k_flatten_main -> k_flatten_tail, what compare_frame just checked):
    # same lines, same image
    import torch

    eng.set_frames_in_flight(3)
    eng.upload_scene(packed, layout)
    targets = [torch.zeros((1600, 1600, 4), dtype=torch.uint8, device="cuda:0") for _ in range(3)]
    torch.cuda.synchronize()
    for t in targets:
        eng.render_resident(1600, 1600, WHITE, AaConfig.Msaa16, out=t)
    assert eng.sync() == 0
    assert eng.bump()["lines"] == bump["lines"]
    for t in targets:
        assert np.array_equal(t.cpu().numpy(), img)


def test_config_c4_mmark_reduced(gpu_engine):
    # mmark generator at a size the oracle finishes quickly (5k elements); binning stress at 2048^2
    packed, layout = workloads.mmark_scene(n=5000).resolve()
    compare_frame(gpu_engine, packed, layout, 2048, 2048, WHITE, AaConfig.Msaa16, "gpu_mmark5k")


def test_config_c4_mmark_50k_full_size(gpu_engine):
    # BASELINE config C4 at full size: 50 000 mmark elements, 2048x2048, MSAA16 (64 bins)
    packed, layout = workloads.mmark_scene().resolve()
    img, ref, bump = compare_frame(gpu_engine, packed, layout, 2048, 2048, WHITE, AaConfig.Msaa16, "gpu_mmark50k")
    assert bump["failed"] == 0


def test_auto_grow_beyond_reference_pools(built):
    # a scene that overflows the reference's fixed 2^21-line pool renders after the robust rounds
    import vello_amd
    from oracle.oracle import Oracle

    packed, layout = workloads.paris_like_scene(seed=0x5EED0011, n_paths=42000).resolve()
    eng = vello_amd.Engine()
    img, bump = eng.render(packed, layout, 1600, 1600, WHITE, AaConfig.Msaa16)
    assert bump["failed"] != 0 and bump["lines"] > (1 << 21)
    eng.set_auto_grow(True)
    img, bump = eng.render(packed, layout, 1600, 1600, WHITE, AaConfig.Msaa16)
    assert bump["failed"] == 0 and eng.capacities()["lines"] > (1 << 21)
    o = Oracle(capacity_scale=2)
    o.set_scene(packed, layout, 1600, 1600, WHITE, int(AaConfig.Msaa16))
    assert np.array_equal(img, o.render())


def test_frames_in_flight_match_oracle(built):
    # three frames of one resident scene in flight on three lanes, each with its own target and its own
    # params (size / base colour / AA): every one must equal the oracle's frame for those params
    import torch
    import vello_amd
    from oracle.oracle import Oracle

    packed, layout = workloads.random_test_scene(5, n_paths=500, size=384.0, strokes=True, clips=False).resolve()
    eng = vello_amd.Engine()
    eng.set_frames_in_flight(3)
    eng.upload_scene(packed, layout)
    variants = [(384, 384, BLACK, AaConfig.Msaa16), (320, 256, WHITE, AaConfig.Msaa8), (384, 200, 0xFF204060, AaConfig.Msaa16)]
    targets = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for (w, h, _, _) in variants]
    torch.cuda.synchronize()
    for rep in range(4):
        for (w, h, base, aa), t in zip(variants, targets):
            eng.render_resident(w, h, base, aa, out=t)
    eng.sync_frame(2)
    assert eng.sync() == 0
    o = Oracle()
    for (w, h, base, aa), t in zip(variants, targets):
        o.set_scene(packed, layout, w, h, base, int(aa))
        assert np.array_equal(o.render(), t.cpu().numpy()), (w, h, hex(base), int(aa))
    # growing the ring after the scene is resident allocates the new lanes' scene buffers as well
    eng.set_frames_in_flight(4)
    for i in range(8):
        (w, h, base, aa), t = variants[i % 3], targets[i % 3]
        eng.render_resident(w, h, base, aa, out=t)
    assert eng.sync() == 0
    for (w, h, base, aa), t in zip(variants, targets):
        o.set_scene(packed, layout, w, h, base, int(aa))
        assert np.array_equal(o.render(), t.cpu().numpy())


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16])
def test_gradient_image_blur_brushes(gpu_engine, aa):
    # gradients (all kinds x extend modes), images (3 qualities, extend modes, BGRA, premultiplied)
    # and blurred rounded rects; exp/pow in the blur follow the fp64-rounded-once rule on both sides
    import vello_amd

    r = vello_amd.Resolver().resolve(workloads.brushes_scene())
    compare_frame(gpu_engine, r.packed, r.layout, 256, 256, WHITE, aa, f"gpu_brushes_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                  resolved=r)


def test_brushes_at_scale(gpu_engine):
    # the same brushes under a 3x zoom and rotation: larger sampled areas, bicubic / bilinear minification paths
    import vello_amd
    from vello_amd import Affine, Scene

    base = workloads.brushes_scene()
    s = Scene()
    s.append(base, Affine.translate(60, -40) * Affine.rotate(0.15) * Affine.scale(3.0))
    r = vello_amd.Resolver().resolve(s)
    compare_frame(gpu_engine, r.packed, r.layout, 800, 800, BLACK, AaConfig.Msaa16, "gpu_brushes_zoom", resolved=r)


def test_flatten_staging_overflow(gpu_engine):
    # k_flatten's LDS staging area holds 3072 lines per workgroup; this scene puts 3728 into one workgroup
    packed, layout = workloads.heavy_strokes_scene().resolve()
    img, ref, bump = compare_frame(gpu_engine, packed, layout, 1024, 1024, BLACK, AaConfig.Msaa16, "gpu_heavy_strokes")
    assert bump["lines"] > 3072


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa16])
@pytest.mark.parametrize("which", ["tricky_strokes", "fill_types", "robust_paths", "funky_paths", "cardioid", "many_draw_objects"])
def test_reference_test_scenes(gpu_engine, which, aa):
    # examples/scenes/src/test_scenes.rs: cusps / 180-degree turns / degenerate cubics under the stroker (:513-697),
    # self-intersections under both fill rules (:699-770), edges exactly on tile boundaries (:1610-1691), path-encoder
    # edge cases (:293-333), 600 long chords in one stroked path (:1306-1331), 90 000 draw objects (:1928-1948)
    scene, w, h = getattr(workloads, which + "_scene")()
    packed, layout = scene.resolve()
    compare_frame(gpu_engine, packed, layout, w, h, BLACK, aa, f"gpu_{which}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0)


def test_tricky_strokes_round_joins_and_caps(gpu_engine):
    # the same cubics with round joins and caps: every arc goes through the fp64 sin / cos / acos / atan2 paths
    from vello_amd import Cap, Join

    scene, w, h = workloads.tricky_strokes_scene(join=Join.Round, cap=Cap.Round)
    packed, layout = scene.resolve()
    compare_frame(gpu_engine, packed, layout, w, h, WHITE, AaConfig.Msaa8, "gpu_tricky_round")


def test_config_c5_all_eight_seeds(built):
    # BASELINE config C5 as bench.py --gpus 8 renders it (bench.py Workload: rank k draws paris_like_scene_d2(SEED0 + k) on a
    # context created with D2_CAPS): the seven scenes of ranks 1..7, images bit-exact against the oracle (tile-parallel);
    # seeds ...0004 and ...0007 with every intermediate as well.  Seed ...0001 (rank 0) is test_config_c3_d2_scene_full_size.
    import bench
    import vello_amd
    from oracle.oracle import Oracle

    eng = vello_amd.Engine(device=0, capacities=bench.D2_CAPS)
    o = Oracle(capacity_scale=8)
    o.set_threads(32)
    for k in range(1, 8):
        packed, layout = workloads.paris_like_scene_d2(bench.SEED0 + k).resolve()
        img, ref, bump = compare_frame(eng, packed, layout, 1600, 1600, WHITE, AaConfig.Msaa16, f"gpu_paris_d2_seed{k}",
                                       check_stages=(k in (3, 6)), oracle=o)
        assert bump["failed"] == 0 and bump["lines"] > (1 << 21)
        assert np.array_equal(img, ref), k


def test_config_c5_r1mix_seeds(gpu_engine):
    # round 1's stroke-light mix (bench.py --workload r1mix) at the same eight seeds, images against the oracle
    from oracle.oracle import Oracle

    o = Oracle()
    o.set_threads(32)
    for k in range(1, 8):  # seed ...0001 is test_config_c3_paris_like_full_size
        packed, layout = workloads.paris_like_scene(0x5EED0001 + k).resolve()
        img, ref, bump = compare_frame(gpu_engine, packed, layout, 1600, 1600, WHITE, AaConfig.Msaa16, f"gpu_paris_seed{k}",
                                       check_stages=False, oracle=o)
        assert bump["failed"] == 0


def test_render_frame_pipelines_scenes(built):
    # animation form: a different scene per frame, 3 frames in flight, targets checked after the whole burst
    import torch
    import vello_amd
    from oracle.oracle import Oracle

    eng = vello_amd.Engine()
    eng.set_frames_in_flight(3)
    scenes = [workloads.random_test_scene(10 + k, n_paths=300, size=320.0, strokes=True, clips=(k % 2 == 1)).resolve() for k in range(3)]
    targets = [torch.zeros((320, 320, 4), dtype=torch.uint8, device="cuda:0") for _ in range(3)]
    torch.cuda.synchronize()
    for rep in range(3):
        for k in range(3):
            eng.render_frame(scenes[k][0], scenes[k][1], 320, 320, BLACK, AaConfig.Msaa16, out=targets[k])
    assert eng.sync() == 0
    o = Oracle()
    for k in range(3):
        o.set_scene(scenes[k][0], scenes[k][1], 320, 320, BLACK, int(AaConfig.Msaa16))
        assert np.array_equal(o.render(), targets[k].cpu().numpy()), k


def test_pure_c_client_of_the_abi(built, tmp_path):
    # a C program (gcc, no Python / C++ in the process) drives include/vello_hip.h: one-shot render and the animation form
    import struct
    import subprocess
    import vello_amd
    from oracle.oracle import Oracle

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.dirname(vello_amd.library_path())
    exe = str(tmp_path / "render_blob")
    subprocess.run(["gcc", "-O1", "-std=c11", "-o", exe, os.path.join(root, "tests", "c_abi", "render_blob.c"), "-L" + lib_dir,
                    "-lvello_hip", "-Wl,-rpath," + lib_dir], check=True)
    r = vello_amd.Resolver().resolve(workloads.brushes_scene())
    w, h, aa = 256, 256, int(AaConfig.Msaa16)
    blob = struct.pack("<10I", *r.layout) + struct.pack("<5I", w, h, WHITE, aa, r.ramps.size // 512) + struct.pack("<Q", r.packed.nbytes)
    blob += r.packed.tobytes() + r.ramps.tobytes()
    (tmp_path / "scene.bin").write_bytes(blob)
    out_path = tmp_path / "out.rgba"
    p = subprocess.run([exe, str(tmp_path / "scene.bin"), str(out_path)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    img = np.frombuffer(out_path.read_bytes(), dtype=np.uint8).reshape(h, w, 4)
    o = Oracle()
    o.set_scene(r.packed, r.layout, w, h, WHITE, aa)
    o.set_ramps(r.ramps)            # no image atlas in the C client: image brushes sample transparent black on both sides
    assert np.array_equal(img, o.render())


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa16])
@pytest.mark.parametrize("which", ["gradient_extend", "blend_grid", "deep_blend", "many_clips", "blurred_rounded_rect", "image_sampling",
                                   "image_sampling_bicubic"])
def test_reference_brush_and_layer_scenes(gpu_engine, which, aa):
    # test_scenes.rs:978-1043 (gradient kinds x extend modes), :1213-1239 + :1398-1436 (16 mix modes over gradients in
    # nested layers), :1241-1276 (blend stack deeper than 4), :1278-1304 (600 clip layers), :1988-2031 (blurred rounded
    # rects), :2053-2113 (image sampling: nearest / bilinear / bicubic under rotation, skew, non-uniform scale)
    import vello_amd

    scene, w, h = getattr(workloads, which + "_scene")()
    r = vello_amd.Resolver().resolve(scene)
    compare_frame(gpu_engine, r.packed, r.layout, w, h, WHITE, aa, f"gpu_{which}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                  resolved=r)


@pytest.mark.parametrize("aa", [AaConfig.Area, AaConfig.Msaa16])
@pytest.mark.parametrize("which", ["ref_stroke_styles", "ref_stroke_styles_non_uniform", "ref_stroke_styles_skew", "two_point_radial",
                                   "conflation_artifacts", "labyrinth", "clip_test", "luminance_mask", "image_extend_modes",
                                   "image_extend_modes_nearest", "brush_transform"])
def test_reference_catalogue_second_batch(gpu_engine, which, aa):
    # test_scenes.rs:335-511 (cap / join / miter-limit matrix under identity, non-uniform scale and skew), :1045-1211
    # (COLR two-point radial cases x extend modes), :1444-1531 (conflation: shared edges of opposite winding), :1533-1608
    # (labyrinth: 140 overlapping sub-paths in one fill), :1708-1911 (even-odd clip, STROKE-styled clip layer, clipped
    # blend layer), :2214-2289 (luminance mask), :2168-2212 (image brush extend modes, bilinear and nearest)
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
    compare_frame(gpu_engine, r.packed, r.layout, w, h, WHITE, aa, f"gpu_{which}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                  resolved=r)


def test_reference_catalogue_third_batch(gpu_engine):
    # test_scenes.rs:2291-2349 at full size (a 640 x 480 image under a luminance-mask layer; the JPEG asset replaced by a synthetic image
    # of the same role) and :1693-1706 (a translucent square over the base colour, which the reference animates through the hues)
    import vello_amd

    scene, w, h = workloads.image_luminance_mask_scene()
    r = vello_amd.Resolver().resolve(scene)
    for aa in (AaConfig.Area, AaConfig.Msaa16):
        compare_frame(gpu_engine, r.packed, r.layout, w, h, WHITE, aa, f"gpu_image_luminance_mask_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                      resolved=r)
    scene, w, h = workloads.base_color_test_scene()
    r = vello_amd.Resolver().resolve(scene)
    for base in (0xFF3C8CE6, 0xFF20C040, 0xFFFFFFFF, 0x80402010):
        for aa in (AaConfig.Area, AaConfig.Msaa16):
            compare_frame(gpu_engine, r.packed, r.layout, w, h, base, aa, f"gpu_base_color_{base:08x}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                          resolved=r)


def test_image_atlas_residency_sequence(gpu_engine):
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
        compare_frame(gpu_engine, r.packed, r.layout, 200, 80, WHITE, AaConfig.Area, f"gpu_residency_{i}", tol=1, resolved=r)
    assert seen_evict and seen_grow


def test_fuzz_whole_api(gpu_engine):
    # workloads/fuzz.py: seeded random scenes over the whole scene API (every brush kind with degenerate parameters, all
    # mix / compose modes, luminance masks, fill- and stroke-styled clips, zero / hairline / huge stroke widths, skewed,
    # mirrored and near-singular affines, repeated points, geometry far off screen and on tile corners)
    import vello_amd
    from workloads.fuzz import fuzz_scene

    gpu_engine.set_auto_grow(True)
    try:
        for seed in range(0, 300):
            r = vello_amd.Resolver().resolve(fuzz_scene(seed))
            aa = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16][seed % 3]
            base = [0xFF000000, 0xFFFFFFFF, 0x00000000, 0x80FF8040][seed % 4]
            compare_frame(gpu_engine, r.packed, r.layout, 128, 128, base, aa, f"gpu_fuzz_{seed}", tol=1 if aa == AaConfig.Area else 0, resolved=r,
                          order_sensitive=True)
    finally:
        gpu_engine.set_auto_grow(False)


def test_fuzz_target_sizes_and_long_scenes(gpu_engine):
    # the same generator at awkward target sizes (one pixel, one tile, not multiples of 16, several bins) and with up to
    # 700 operations per scene (several 256-draw batches per bin in coarse)
    import vello_amd
    from workloads.fuzz import fuzz_scene

    gpu_engine.set_auto_grow(True)
    try:
        for seed in range(0, 150):
            w, h = [(333, 205), (97, 530), (512, 512), (16, 16), (1, 1), (700, 40)][seed % 6]
            r = vello_amd.Resolver().resolve(fuzz_scene(seed, size=max(w, h, 8), n_ops=[40, 700, 300][seed % 3]))
            aa = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16][(seed // 2) % 3]
            base = [0xFF000000, 0xFFFFFFFF, 0x00000000, 0x80FF8040][seed % 4]
            compare_frame(gpu_engine, r.packed, r.layout, w, h, base, aa, f"gpu_fuzz2_{seed}", tol=1 if aa == AaConfig.Area else 0, resolved=r,
                          order_sensitive=True)
    finally:
        gpu_engine.set_auto_grow(False)


def test_fuzz_auto_grow_from_tiny_pools(built):
    # robust dynamic memory under fuzz: every pool starts at 64 elements, auto-grow has to find the frame's
    # demand stage by stage (a failed stage hides the demand of the later ones) and the final frame must equal the oracle's
    import vello_amd
    from workloads.fuzz import fuzz_scene

    for seed in range(0, 40):
        eng = vello_amd.Engine(capacities={"lines": 64, "binning": 64, "tile": 64, "seg_counts": 64, "segments": 64, "blend": 16,
                                           "ptcl": 64 * 4 + 64})
        eng.set_auto_grow(True)
        w, h = [(128, 128), (300, 200), (64, 64)][seed % 3]
        r = vello_amd.Resolver().resolve(fuzz_scene(seed, size=max(w, h), n_ops=[40, 300][seed % 2]))
        aa = [AaConfig.Area, AaConfig.Msaa8, AaConfig.Msaa16][(seed // 2) % 3]
        compare_frame(eng, r.packed, r.layout, w, h, 0xFF203040, aa, f"gpu_fuzz3_{seed}", tol=1 if aa == AaConfig.Area else 0, resolved=r,
                      order_sensitive=True)
        del eng


def test_inconsistent_scenes_are_refused(gpu_engine):
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
        "more clips in the layout than clip tags": (good_packed, layout._replace(n_clips=layout.n_clips + 3)),
    }
    for what, (packed, lay) in cases.items():
        with pytest.raises(vello_amd.VelloHipError):
            gpu_engine.render(packed, lay, 256, 256, WHITE, AaConfig.Msaa8)
        # the context is not poisoned: the next (valid) frame is right
        compare_frame(gpu_engine, good_packed, layout, 256, 256, WHITE, AaConfig.Msaa8, "gpu_invalid_after_" + what.split(":")[0].replace(" ", "_"))


def test_reference_regression_scenes(gpu_engine):
    # the reference's own regression tests that need no fonts: known_issues.rs:54-90 (clip_blends, issue #1198),
    # regression.rs:18-31 (rounded_rectangle_watertight, issue #616), :107-121 (stroke_width_zero, issue #662), known_issues.rs:20-52
    # (layer_size, issue #1061, `should_panic` upstream: an EMPTY Compose::Clear layer over a red square -- held to the restated shaders' frame)
    import vello_amd

    cases = [workloads.clip_blends_scene() + ("clip_blends",)] + workloads.regression_stroke_scenes()
    for scene, w, h, name in cases:
        r = vello_amd.Resolver().resolve(scene)
        for aa in (AaConfig.Area, AaConfig.Msaa16):
            img, _, _ = compare_frame(gpu_engine, r.packed, r.layout, w, h, BLACK, aa, f"gpu_{name}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0,
                                      resolved=r)
        if name == "stroke_width_zero":
            assert (img[:, :, :3] == 0).all()
        if name == "clip_blends":
            assert tuple(img[5, 5]) == (0, 0, 255, 255) and tuple(img[90, 50]) == (0, 0, 212, 255)   # blue x aquamarine, multiplied
        if name == "layer_size":  # (the Clear layer's rectangle comes out transparent -- over black: 0 -- the green around it stays)
            assert tuple(img[5, 5]) == (0, 255, 0, 255) and tuple(img[30, 30]) == (0, 0, 0, 0)


def test_zero_width_stroke_clip_before_any_transform(gpu_engine):
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
    img, _, _ = compare_frame(gpu_engine, r.packed, r.layout, 80, 80, BLACK, AaConfig.Msaa8, "gpu_zero_width_clip_first", resolved=r)
    assert (img[:, :, 1] == 0).all(), "the zero-width stroke clip suppresses everything inside the layer"
    assert tuple(img[20, 20]) == (255, 0, 0, 255)


def test_large_target_with_auto_grow(built):
    # 6000 x 6000 = 140 625 tiles x 64 PTCL words: past the fixed 2^23-word pool; robust mode sizes the pool for the target
    import vello_amd
    from vello_amd import Affine, Scene
    from oracle.oracle import Oracle

    base = workloads.random_test_scene(21, n_paths=400, size=512.0, strokes=True, clips=False)
    s = Scene()
    s.append(base, Affine.scale(6000.0 / 512.0))
    packed, layout = s.resolve()
    eng = vello_amd.Engine()
    with pytest.raises(vello_amd.VelloHipError):
        eng.render(packed, layout, 6000, 6000, BLACK, AaConfig.Msaa8)
    eng.set_auto_grow(True)
    img, bump = eng.render(packed, layout, 6000, 6000, BLACK, AaConfig.Msaa8)
    assert bump["failed"] == 0 and eng.capacities()["ptcl"] > (1 << 23)
    o = Oracle(capacity_scale=4)
    o.set_threads(32)
    o.set_scene(packed, layout, 6000, 6000, BLACK, int(AaConfig.Msaa8))
    assert np.array_equal(img, o.render())


def test_estimator_gives_one_round_from_tiny_pools(built):
    # vello_hip_estimate_capacities + robust mode: the 42k-path scene of test_auto_grow_beyond_reference_pools
    # and a curve-heavy one render in ONE round from pools of 64 elements
    import vello_amd
    from oracle.oracle import Oracle
    from vello_amd.renderer import estimate_capacities

    tiger = np.load(os.path.join(GOLD, "tiger_scene.npz"))
    for packed, layout, w, h in (workloads.paris_like_scene(n_paths=42000, size=1000.0).resolve() + (1000, 1000),
                                 (tiger["packed"], Layout(*[int(v) for v in tiger["layout"]]), 1024, 1024)):
        o = Oracle(capacity_scale=4)
        o.set_scene(packed, layout, w, h, WHITE, int(AaConfig.Msaa16))
        ref = o.render()
        need = o.bump()
        est = estimate_capacities(packed, layout, w, h)
        assert est["lines"] >= need["lines"] and est["seg_counts"] >= need["seg_counts"] and est["tiles"] >= need["tile"]
        eng = vello_amd.Engine(device=0, capacities={"lines": 64, "seg_counts": 64, "segments": 64, "tiles": 64,
                                                     "bin_data": layout.bin_data_start + 64, "ptcl": 64 * ((w + 15) // 16) * ((h + 15) // 16) + 512})
        eng.set_auto_grow(True)
        img, bump = eng.render(packed, layout, w, h, WHITE, AaConfig.Msaa16)
        assert bump["failed"] == 0 and np.array_equal(img, ref)
        assert eng.last_render_attempts() == 1


def test_gather_frames_peer_copy(built):
    # vello_hip_gather_frames on hardware: two contexts on GPU 0 (the single-GPU form of the per-GPU contexts of a
    # one-process host), hipMemcpyPeerAsync on per-context copy streams ordered behind the frames by events
    import torch
    import vello_amd
    from oracle.oracle import Oracle
    from vello_amd.renderer import gather_frames

    scenes = [workloads.stroke_styles_scene(), workloads.clip_blend_scene(), workloads.random_test_scene(5, n_paths=200, size=256.0)]
    engines, srcs, refs = [], [], []
    dst = torch.zeros((3, 256, 256, 4), dtype=torch.uint8, device="cuda:0")
    for sc in scenes:
        packed, layout = sc.resolve()
        e = vello_amd.Engine(device=0)
        e.upload_scene(packed, layout)
        src = torch.zeros((256, 256, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()  # torch's zero fills (torch's stream) before the engine's streams touch the buffers
        e.render_resident(256, 256, BLACK, AaConfig.Msaa16, out=src)   # NOT waited for: the gather orders itself behind it
        o = Oracle()
        o.set_scene(packed, layout, 256, 256, BLACK, int(AaConfig.Msaa16))
        refs.append(o.render())
        engines.append(e)
        srcs.append(src)
    gather_frames(engines, srcs, [dst[i] for i in range(3)], 256 * 256 * 4)
    out = dst.cpu().numpy()
    for i in range(3):
        assert np.array_equal(out[i], refs[i]), i


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_gpu_fill_rules_interleaved_in_one_tile(gpu_engine, seed):
    # fine's three MSAA resolve paths (sparse non-zero, dense even-odd, unbatched) handing the sample counters to each other
    from tests.test_emu_parity import _fill_rule_interleave_scene

    packed, layout = _fill_rule_interleave_scene(seed)
    for aa in (AaConfig.Msaa8, AaConfig.Msaa16):
        compare_frame(gpu_engine, packed, layout, 32, 32, 0xFF203040, aa, f"gpu_rules_{seed}_{int(aa)}", order_sensitive=True)


@pytest.mark.gpu
def test_gpu_path_count_both_forms(gpu_engine):
    from tests.test_emu_parity import path_count_both_forms

    path_count_both_forms(gpu_engine, "gpu_path_count_forms")


@pytest.mark.gpu
def test_gpu_stroke_kernel_untame_inputs(gpu_engine):
    from tests.test_emu_parity import stroke_kernel_untame_inputs

    stroke_kernel_untame_inputs(gpu_engine, "gpu_stroke_untame", seeds=(4552, 8707, 11797, 11851))


@pytest.mark.gpu
def test_gpu_path_count_long_lines(gpu_engine):
    from tests.test_emu_parity import path_count_long_lines

    path_count_long_lines(gpu_engine, "gpu_pc_long")


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(8))
def test_gpu_stroked_line_kernel(gpu_engine, case):
    # flatten's stroked-line kernel (normally from 393 216 stroked lines on: the d2 scene) forced on the stroke catalogue
    from tests.test_emu_parity import _stroke_kernel_cases

    name, packed, layout, w, h = _stroke_kernel_cases()[case]
    gpu_engine.set_debug_flags(stroke_kernel=True)
    gpu_engine.set_auto_grow(True)
    try:
        for aa in (AaConfig.Area, AaConfig.Msaa16):
            compare_frame(gpu_engine, packed, layout, w, h, 0xFFFFFFFF, aa, f"gpu_strokekernel_{name}_{int(aa)}", tol=1 if aa == AaConfig.Area else 0)
        # with frames in flight the stroke workgroups are a kernel of their own ahead of the heavy list's (k_flatten_strokes,
        # k_flatten_heavy) instead of part of its launch (k_flatten_main, k_flatten_tail)
        gpu_engine.set_frames_in_flight(2)
        compare_frame(gpu_engine, packed, layout, w, h, 0xFFFFFFFF, AaConfig.Msaa16, f"gpu_strokekernel_{name}_2inflight")
    finally:
        gpu_engine.set_frames_in_flight(1)
        gpu_engine.set_debug_flags()
        gpu_engine.set_auto_grow(False)


@pytest.mark.parametrize("case", range(8))
def test_gpu_flatten_kernel_sets(gpu_engine, case):
    # flatten's two sets of heavy-list kernels (flatten_walk.inc), forced, on the stroke catalogue
    from tests.test_emu_parity import _stroke_kernel_cases, flatten_kernel_sets

    name, packed, layout, w, h = _stroke_kernel_cases()[case]
    gpu_engine.set_auto_grow(True)
    try:
        flatten_kernel_sets(gpu_engine, "gpu_flsets_" + name, packed, layout, w, h)
    finally:
        gpu_engine.set_auto_grow(False)


@pytest.mark.parametrize("case", range(7))
def test_gpu_front_fusion(gpu_engine, case):
    # small scenes: the workgroups of consecutive stages as turns of one launch with a grid barrier between the stages (k_front);
    # the launches counted, every stage and the image against the oracle, one frame in flight and two; then unfused
    from tests.parity import check_front_fusion, front_fusion_cases

    check_front_fusion(gpu_engine, front_fusion_cases()[case])


def test_gpu_front_fusion_barrier_stress():
    # k_front's grid barrier under load: the tiger (16 workgroups a launch, on different XCDs) 600 times with four frames in
    # flight, and a scene whose every stage is one workgroup's -- each frame identical to the oracle's
    import torch
    import vello_amd
    from oracle.oracle import Oracle

    d = np.load(os.path.join(GOLD, "tiger_scene.npz"))
    layout = Layout(*[int(v) for v in d["layout"]])
    for packed, lay, w, h, per_frame in ((d["packed"], layout, 512, 512, 2), tuple(workloads.circle_scene().resolve()) + (256, 256, 1)):
        o = Oracle()
        o.set_scene(packed, lay, w, h, WHITE, int(AaConfig.Msaa8))
        ref = o.render()
        eng = vello_amd.Engine()
        eng.set_frames_in_flight(4)
        eng.upload_scene(packed, lay)
        targets = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
        torch.cuda.synchronize()
        before = eng.fused_launches()
        for rep in range(150):
            for t in targets:
                eng.render_resident(w, h, WHITE, AaConfig.Msaa8, out=t)
            if rep % 50 == 49:
                assert eng.sync() == 0
                for t in targets:
                    assert np.array_equal(ref, t.cpu().numpy()), rep
                    t.zero_()
                torch.cuda.synchronize()
        assert eng.fused_launches() - before == 600 * per_frame


def test_gpu_flatten_kernel_sets_baseline_configs(gpu_engine):
    # ... and on BASELINE's C2 (tiger: a wave per curve) and C4 (mmark-50k: a list of stroked curves), whichever set the engine
    # would pick for them by itself
    from tests.test_emu_parity import flatten_kernel_sets

    d = np.load(os.path.join(GOLD, "tiger_scene.npz"))
    flatten_kernel_sets(gpu_engine, "gpu_flsets_tiger", d["packed"], Layout(*[int(v) for v in d["layout"]]), 1024, 1024)
    packed, layout = workloads.mmark_scene().resolve()
    gpu_engine.set_auto_grow(True)
    try:
        flatten_kernel_sets(gpu_engine, "gpu_flsets_mmark", packed, layout, 2048, 2048, in_flight=False)
    finally:
        gpu_engine.set_auto_grow(False)


def test_clip_stage_partitioned(gpu_engine):
    # a5: clip_reduce / clip_leaf as partitioned kernels (clip.hip: 256 clips per workgroup, one workgroup over the partitions, a
    # second pass per partition) and as the one-wave stack machine, against the oracle's sequential stack; up to 300 000 clips and
    # 20 000 open layers (the reference's kernels stop at 65 536 clips / 256 open layers, clip_leaf.wgsl:87-112)
    from oracle.oracle import Oracle
    from tests.parity import clip_structures, compare_clip_stage

    oracle = Oracle()
    for name, ops in clip_structures(big=True):
        compare_clip_stage(gpu_engine, ops, np.random.default_rng(len(ops)), "gpu clips " + name, oracle=oracle)


def test_write_image_is_ordered_with_frames_in_flight(built):
    # vello_hip_write_image does not wait for the frames in flight: a "video texture" whose pixels change before every frame,
    # four frames in flight, nothing waited for until the end -- frame k must show the k-th upload (not the one before it:
    # the frame waits for the transfer; not the one after it: the transfer waits for the frames that still sample the atlas)
    import torch
    import vello_amd
    from vello_amd import Affine, ImageBrush, ImageData, ImageQuality, Scene

    n = 12
    frames = [np.full((64, 64, 4), 255, dtype=np.uint8) for _ in range(n)]
    for k, px in enumerate(frames):
        px[:, :, 0] = 20 * k
        px[:, :, 1] = 255 - 20 * k
        px[k:k + 8, :, 2] = 7
    s = Scene()
    s.draw_image(ImageBrush(ImageData(frames[0]), quality=ImageQuality.Low), Affine.translate(8.0, 8.0) * Affine.scale(6.0))
    r = vello_amd.Resolver().resolve(s)
    (x, y, _), = r.uploads
    eng = vello_amd.Engine()
    eng.set_frames_in_flight(4)
    eng.upload_resolved(r)
    targets = [torch.zeros((400, 400, 4), dtype=torch.uint8, device="cuda:0") for _ in range(n)]
    torch.cuda.synchronize()
    for k in range(n):
        eng.write_image(x, y, frames[k])
        eng.render_resident(400, 400, BLACK, AaConfig.Msaa16, out=targets[k])
    assert eng.sync() == 0
    for k in range(n):
        got = targets[k].cpu().numpy()
        # texel (i, j) covers pixels 8 + 6 i .. 8 + 6 i + 5: sample the centres of a few texels
        for ty in (0, k, k + 7, k + 8 if k + 8 < 64 else 0, 63):
            for tx in (0, 31, 63):
                assert tuple(got[8 + 6 * ty + 3, 8 + 6 * tx + 3]) == tuple(frames[k][ty, tx]), (k, ty, tx)


def test_atlas_clear_is_ordered_with_the_uploads_that_follow(built):
    # vello_hip_resize_image_atlas cleared the new atlas with hipMemset on the null stream while
    # vello_hip_write_image transfers on a non-blocking upload stream -- nothing ordered the upload behind the clear, and a
    # clear that lands late zeroes the texels (the brushes test failed that way on one fresh box).  A large atlas makes the
    # clear long (8192^2 x 4 B = 256 MB) and the upload follows at once: every round must sample the uploaded texels.
    import torch
    import vello_amd
    from vello_amd import Affine, ImageBrush, ImageData, ImageQuality, Scene

    px = np.full((64, 64, 4), 255, dtype=np.uint8)
    px[:, :, 0] = np.arange(64, dtype=np.uint8)[None, :] * 3
    px[:, :, 1] = np.arange(64, dtype=np.uint8)[:, None] * 2 + 9
    s = Scene()
    s.draw_image(ImageBrush(ImageData(px), quality=ImageQuality.Low), Affine.translate(8.0, 8.0) * Affine.scale(6.0))
    r = vello_amd.Resolver().resolve(s)
    (x, y, _), = r.uploads
    eng = vello_amd.Engine()
    eng.upload_scene(r.packed, r.layout, r.ramps)
    target = torch.zeros((400, 400, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    for rnd in range(12):
        side = 8192 if rnd % 2 == 0 else 4096  # alternating sizes: a fresh allocation and a reused one
        eng.resize_image_atlas(side, side)
        eng.write_image(x, y, px)
        eng.render_resident(400, 400, BLACK, AaConfig.Msaa16, out=target)
        assert eng.sync() == 0
        got = target.cpu().numpy()
        for ty in (0, 17, 63):
            for tx in (0, 31, 63):
                assert tuple(got[8 + 6 * ty + 3, 8 + 6 * tx + 3]) == tuple(px[ty, tx]), (rnd, ty, tx)
    # texels never uploaded read as transparent black: the clear itself happened
    eng.resize_image_atlas(4096, 4096)
    eng.render_resident(400, 400, BLACK, AaConfig.Msaa16, out=target)
    assert eng.sync() == 0
    got = target.cpu().numpy()
    assert tuple(got[8 + 6 * 17 + 3, 8 + 6 * 31 + 3]) == (0, 0, 0, 255)


def _slice_cases():
    import vello_amd

    def tiger(e):
        d = np.load(os.path.join(GOLD, "tiger_scene.npz"))
        compare_frame(e, d["packed"], Layout(*[int(v) for v in d["layout"]]), 1024, 1024, WHITE, AaConfig.Msaa8, "gpu_tiger_slices")

    def random_clips(e):
        packed, layout = workloads.random_test_scene(2, n_paths=1500, size=768.0, strokes=True, clips=True).resolve()
        compare_frame(e, packed, layout, 768, 768, BLACK, AaConfig.Msaa16, "gpu_random_slices")

    def brushes(e):
        r = vello_amd.Resolver().resolve(workloads.brushes_scene())
        compare_frame(e, r.packed, r.layout, 256, 256, WHITE, AaConfig.Msaa16, "gpu_brushes_slices", resolved=r)

    def clip_blend(e):
        packed, layout = workloads.clip_blend_scene().resolve()
        compare_frame(e, packed, layout, 256, 256, BLACK, AaConfig.Msaa8, "gpu_clips_slices")

    def r1mix(e):
        packed, layout = workloads.paris_like_scene().resolve()
        compare_frame(e, packed, layout, 1600, 1600, WHITE, AaConfig.Msaa16, "gpu_paris_slices", check_stages=False, back_half=False)

    return [("tiger", tiger), ("random_clips", random_clips), ("brushes", brushes), ("clip_blend", clip_blend), ("r1mix", r1mix)]


@pytest.mark.parametrize("name,body", _slice_cases(), ids=[c[0] for c in _slice_cases()])
def test_fine_slices_forced(gpu_engine, name, body):
    # fine's sliced path on the MI355X with EVERY tile cut into slices of 4 fills (VELLO_HIP_DEBUG_FINE_SLICES): the slices of a
    # tile run on whatever CUs / XCDs the dispatcher picks, the last one to finish composites from the coverage scratch --
    # the cross-workgroup hand-off the emulator cannot say anything about.  Images bit-exact against the oracle.
    gpu_engine.set_debug_flags(fine_slices=True)
    try:
        body(gpu_engine)
        assert name == "brushes" or gpu_engine.fine_slice_stats()[0] > 0, "no tile was cut into slices"  # (brushes: < 5 fills per tile)
    finally:
        gpu_engine.set_debug_flags()


def test_fine_slices_repeatable_across_frames_in_flight(built):
    # 4 frames in flight, all tiles sliced: every lane has its own slice items / counters / coverage scratch, and 12 frames of
    # the same scene must be the same image (a stale counter or a scratch shared between lanes would show)
    import torch
    import vello_amd

    packed, layout = workloads.random_test_scene(5, n_paths=1200, size=640.0, strokes=True, clips=True).resolve()
    eng = vello_amd.Engine()
    eng.set_debug_flags(fine_slices=True)
    eng.set_frames_in_flight(4)
    eng.upload_scene(packed, layout)
    targets = [torch.zeros((640, 640, 4), dtype=torch.uint8, device="cuda:0") for _ in range(12)]
    torch.cuda.synchronize()
    for t in targets:
        eng.render_resident(640, 640, BLACK, AaConfig.Msaa16, out=t)
    assert eng.sync() == 0
    first = targets[0].cpu().numpy()
    eng.set_debug_flags()
    ref = torch.zeros((640, 640, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    eng.render_resident(640, 640, BLACK, AaConfig.Msaa16, out=ref)
    assert eng.sync() == 0
    assert np.array_equal(first, ref.cpu().numpy())
    for t in targets[1:]:
        assert np.array_equal(first, t.cpu().numpy())


def test_fine_slice_handoff_stress_across_xcds(built):
    # The slice -> compositor hand-off of k_fine (fine.hip: write-through sc1 stores, s_waitcnt vmcnt(0), a relaxed agent-scope
    # ticket, agent-scope loads in the winning wave) is outside what the HSA memory model promises and is pinned to gfx950 by a
    # compile-time guard.  This is the stress half of that bargain: workgroup b of a launch runs on XCD
    # b mod 8 (profiles/r04_atomic_scope.txt: HW_REG_XCC_ID, 4 096 of 4 096) and the slices of a tile are CONSECUTIVE workgroups
    # of k_fine's grid, so with every tile cut into slices of 4 fills each tile's slices sit on different XCDs -- different L2s --
    # by construction.  The road-map scene (thousands of sliced tiles per frame), four frames in flight so that other kernels'
    # traffic shares the caches, 64 frames: every frame must equal the unsliced image bit for bit.
    import torch
    import vello_amd

    packed, layout = workloads.paris_like_scene().resolve()
    eng = vello_amd.Engine()
    eng.upload_scene(packed, layout)
    ref = torch.zeros((1600, 1600, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    eng.render_resident(1600, 1600, WHITE, AaConfig.Msaa16, out=ref)
    assert eng.sync() == 0
    ref_np = ref.cpu().numpy()
    eng.set_debug_flags(fine_slices=True)
    eng.set_frames_in_flight(4)
    targets = [torch.zeros((1600, 1600, 4), dtype=torch.uint8, device="cuda:0") for _ in range(8)]
    torch.cuda.synchronize()
    for rnd in range(8):
        for t in targets:
            eng.render_resident(1600, 1600, WHITE, AaConfig.Msaa16, out=t)
        assert eng.sync() == 0
        # (every sliced tile has >= 2 slice items -- a tile is cut only from FINE_SLICE_MIN_FILLS_FORCED = 5 fills on, into slices
        # of 4 -- so >= 2 XCDs a tile; r1mix: 13 fills a tile on average, thousands of items a frame)
        assert eng.fine_slice_stats()[0] >= 8000, eng.fine_slice_stats()
        for i, t in enumerate(targets):
            assert np.array_equal(ref_np, t.cpu().numpy()), f"round {rnd}, frame {i}: a sliced tile differs from the unsliced image"
    eng.set_debug_flags()


def test_kernel_ms_splits_the_stages_of_several_kernels(built):
    # vello_hip_get_kernel_ms: flatten (light / strokes / heavy) and coarse (prep / coarse) timed kernel by kernel with events
    # between the launches; the parts add up to the stage (they share its first and last event)
    import vello_amd
    from vello_amd.renderer import STAGES

    packed, layout = workloads.random_test_scene(9, n_paths=2000, size=1024.0, strokes=True, clips=True).resolve()
    eng = vello_amd.Engine()
    eng.upload_scene(packed, layout)
    for _ in range(3):
        eng.render_resident(1024, 1024, BLACK, AaConfig.Msaa16)
        eng.sync()
    eng.set_profiling(STAGES)
    eng.stage_ms(); eng.kernel_ms()
    n = 8
    for _ in range(n):
        eng.render_resident(1024, 1024, BLACK, AaConfig.Msaa16)
        eng.sync_frame(0)
    st, km = eng.stage_ms(), eng.kernel_ms()
    eng.set_profiling([])
    for stage, names in eng.KERNELS.items():
        assert st[stage][1] == n and all(km[k][1] == n for k in names)
        parts = sum(km[k][0] for k in names)
        assert parts > 0.0 and abs(parts - st[stage][0]) <= 0.02 * st[stage][0] + 0.01, (stage, parts, st[stage][0])
    assert km["k_flatten_light"][0] > 0.0 and km["k_coarse"][0] > km["k_coarse_prep"][0] * 0.1
    # read-and-reset
    assert all(v == (0.0, 0) for v in eng.kernel_ms().values())
