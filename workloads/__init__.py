"""Workload definitions for tests and bench.py: the BASELINE.json configs restated as
concrete, seeded, synthetic inputs.  Harness code, not a product feature.

  C1  circle_scene            single filled Circle, 256x256, Area           (README example colour)
  C2  tiger_scene             Ghostscript_Tiger.svg via a pico_svg-equivalent loader, 1024x1024, MSAA8
  C3  paris_like_scene        SYNTHETIC stand-in for paris-30k (the SVG is not in the reference tree)
  C4  mmark_scene             port of examples/scenes/src/mmark.rs with a seeded RNG, 50k elements
  C5  8 x C3 with seeds 0x5EED0001..8, one per GPU (bench.py --gpus 8)
"""
from .scenes import (circle_scene, smoke_circle_scene, smoke_square_scene, smoke_gradient_alpha_scene, smoke_data_image_scene, property_image_scene, paris_like_scene, paris_like_scene_d2, mmark_scene,
                     random_test_scene, clip_blend_scene, stroke_styles_scene, brushes_scene, heavy_strokes_scene)
from .pico_svg import load_svg, tiger_scene
from .ref_scenes import (tricky_strokes_scene, fill_types_scene, robust_paths_scene, gradient_extend_scene, blend_grid_scene,
                         deep_blend_scene, many_clips_scene, blurred_rounded_rect_scene, image_sampling_scene,
                         image_sampling_bicubic_scene, funky_paths_scene, cardioid_scene, many_draw_objects_scene,
                         ref_stroke_styles_scene, two_point_radial_scene, conflation_artifacts_scene, labyrinth_scene,
                         clip_test_scene, luminance_mask_scene, image_luminance_mask_scene, base_color_test_scene, image_extend_modes_scene, brush_transform_scene,
                         clip_blends_scene, regression_stroke_scenes)
