"""pico_svg-equivalent SVG loader (examples/scenes/src/pico_svg.rs:45-259, svg.rs:74-104).

Supports what pico_svg supports: <g transform>, <path d fill stroke stroke-width fill-opacity
stroke-opacity opacity>, root viewBox/width/height.  Colours: #rgb, #rrggbb and a few names
(upstream defers to the `color` crate's CSS parser; unknown strings become fuchsia @ 0.5 as upstream).
"""
import xml.etree.ElementTree as ET

from vello_amd import Affine, BezPath, Color, Fill, Scene, Stroke

_NAMED = {"black": (0, 0, 0), "white": (255, 255, 255), "red": (255, 0, 0), "lime": (0, 255, 0), "green": (0, 128, 0),
          "blue": (0, 0, 255), "yellow": (255, 255, 0), "gray": (128, 128, 128), "grey": (128, 128, 128),
          "fuchsia": (255, 0, 255), "none": None}


def parse_color(s):
    s = s.strip()
    if s.startswith("#"):
        h = s[1:]
        if len(h) == 3:
            return Color.from_rgb8(*[int(ch * 2, 16) for ch in h])
        if len(h) == 6:
            return Color.from_rgb8(int(h[0:2], 16), int(h[2:4], 16), int(h[4:6], 16))
    if s.lower() in _NAMED and _NAMED[s.lower()] is not None:
        return Color.from_rgb8(*_NAMED[s.lower()])
    return Color.from_rgb8(255, 0, 255).with_alpha(0.5)


def _opacity(color, attr, node):
    v = node.get(attr)
    if v is None:
        return color
    try:
        a = float(v[:-1]) * 0.01 if v.endswith("%") else float(v)
    except ValueError:
        a = 1.0
    return color.with_alpha(min(max(a, 0.0), 1.0))


def parse_transform(t):
    nt = Affine.IDENTITY
    for ts in [x.strip() for x in t.split(")")]:
        if ts.startswith("matrix("):
            vals = [float(v) for v in ts[7:].replace(",", " ").split()]
            nt = nt * Affine(vals)
        elif ts.startswith("translate("):
            vals = [float(v) for v in ts[10:].replace(",", " ").split()]
            if len(vals) == 2:
                nt = nt * Affine.translate(vals[0], vals[1])
        elif ts.startswith("scale("):
            vals = [float(v) for v in ts[6:].replace(",", " ").split()]
            if len(vals) == 2:
                nt = nt * Affine.scale_non_uniform(vals[0], vals[1])
            elif len(vals) == 1:
                nt = nt * Affine.scale(vals[0])
    return nt


def _local(tag):
    return tag.split("}")[-1]


def _rec(node, fill, scale):
    """Returns a Scene for `node`'s subtree (render_svg_rec: groups are appended with their affine)."""
    scene = Scene()
    for child in node:
        f = fill
        fc = child.get("fill")
        if fc is not None:
            if fc == "none":
                f = None
            else:
                f = _opacity(_opacity(parse_color(fc), "fill-opacity", child), "opacity", child)
        name = _local(child.tag)
        if name == "g":
            aff = parse_transform(child.get("transform")) if child.get("transform") else Affine.IDENTITY
            scene.append(_rec(child, f, scale), aff)
        elif name == "path":
            path = BezPath.from_svg(child.get("d"))
            if f is not None:
                scene.fill(Fill.NonZero, Affine.IDENTITY, f, None, path)
            sc = child.get("stroke")
            if sc is not None and sc != "none":
                try:
                    w = float(child.get("stroke-width", "1"))
                except ValueError:
                    w = 1.0
                w *= abs(scale)
                col = _opacity(_opacity(parse_color(sc), "stroke-opacity", child), "opacity", child)
                scene.stroke(Stroke(w), Affine.IDENTITY, col, None, path)
    return scene


def load_svg(xml_string, scale=1.0):
    """PicoSvg::load + render_svg_rec -> (Scene, (width, height))."""
    root = ET.fromstring(xml_string)
    width = float(root.get("width")) if root.get("width") else None
    height = float(root.get("height")) if root.get("height") else None
    vb = root.get("viewBox")
    origin = vsize = None
    if vb:
        v = [float(x) for x in vb.split(" ")]
        if len(v) == 4:
            origin, vsize = (v[0], v[1]), (v[2], v[3])
    transform = Affine.translate(-origin[0], -origin[1]) if origin else Affine.IDENTITY
    if vsize and width and height:
        transform = transform * Affine.scale_non_uniform(width / vsize[0], height / vsize[1])
    elif vsize and width:
        transform = transform * Affine.scale(width / vsize[0])
    elif vsize and height:
        transform = transform * Affine.scale(height / vsize[1])
    if vsize and not width and not height:
        size = vsize
    elif not vsize:
        size = (width or 300.0, height or 150.0)
    else:
        size = (width or vsize[0], height or vsize[1])
    transform = transform * (Affine.scale(scale) if scale >= 0 else Affine((-scale, 0, 0, scale, 0, 0)))
    inner = _rec(root, Color.from_rgb8(0, 0, 0), scale)
    scene = Scene()
    scene.append(inner, transform)
    return scene, size


def tiger_scene(svg_text, width, height):
    """encode_test_scene-style fit (vello_tests/src/lib.rs:293-299): scale to fill the target."""
    inner, size = load_svg(svg_text, 1.0)
    s = min(width / size[0], height / size[1])
    outer = Scene()
    outer.append(inner, Affine.scale(s))
    return outer
