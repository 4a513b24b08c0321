"""Generates the committed fixtures under tests/golden/ from the read-only reference checkout.

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
The GPU box has no /root/reference; tests only read the generated .npz files.

  smoke_goldens.npz   decoded vello_tests/snapshots/smoke/filled_{circle,square}.png (the only real,
                      non-LFS reference snapshots of solid-fill scenes), plus the three other
                      real smoke snapshots that need no fonts: gradient_color_alpha_{premultiplied,unpremultiplied}.png
                      (regression.rs:150-209) and data_image_roundtrip.png (regression.rs:33-104; RGB as the reference
                      compares it, and RGBA as `image` decodes it to feed the scene).  layer_size.png is the snapshot of
                      a test the reference itself is known to fail (known_issues.rs:21-52, #[should_panic]); the two
                      glyph snapshots need a font stack.  None of those three is a pin.
  tiger_scene.npz     Ghostscript_Tiger.svg (examples/assets) encoded through the pico_svg-equivalent
                      loader: packed scene bytes + Layout for a 1024x1024 fit (BASELINE config C2)
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
REF = "/root/reference"


def main():
    circle = np.array(Image.open(f"{REF}/vello_tests/snapshots/smoke/filled_circle.png").convert("RGB"))
    square = np.array(Image.open(f"{REF}/vello_tests/snapshots/smoke/filled_square.png").convert("RGB"))
    smoke = f"{REF}/vello_tests/snapshots/smoke"
    grad_pre = np.array(Image.open(f"{smoke}/gradient_color_alpha_premultiplied.png").convert("RGB"))
    grad_unpre = np.array(Image.open(f"{smoke}/gradient_color_alpha_unpremultiplied.png").convert("RGB"))
    data_image = Image.open(f"{smoke}/data_image_roundtrip.png")
    np.savez_compressed(os.path.join(HERE, "smoke_goldens.npz"), filled_circle=circle, filled_square=square,
                        gradient_color_alpha_premultiplied=grad_pre, gradient_color_alpha_unpremultiplied=grad_unpre,
                        data_image_roundtrip_rgb=np.array(data_image.convert("RGB")),
                        data_image_roundtrip_rgba=np.array(data_image.convert("RGBA")))
    import workloads

    svg = open(f"{REF}/examples/assets/Ghostscript_Tiger.svg").read()
    scene = workloads.tiger_scene(svg, 1024, 1024)
    packed, layout = scene.resolve()
    np.savez_compressed(os.path.join(HERE, "tiger_scene.npz"), packed=packed, layout=np.array(layout, dtype=np.uint32))
    print("smoke goldens", circle.shape, square.shape, "tiger packed", packed.nbytes, "layout", tuple(layout))


if __name__ == "__main__":
    main()
