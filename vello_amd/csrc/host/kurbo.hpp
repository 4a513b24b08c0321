// Minimal subset of kurbo (third-party, NOT vendored under /root/reference; pinned there as
// kurbo 0.13.1 via peniko 0.6.1, Cargo.toml:104-106) needed to feed vello::Scene:
// Point/Affine/PathEl/BezPath, Shape::path_elements for Rect/Circle/Line/RoundedRect,
// Arc::append_iter, BezPath::from_svg, Stroke.  Call sites this serves:
// vello_encoding/src/path.rs:655-657 (Shape::path_elements(0.1)),
// examples/scenes/src/pico_svg.rs:167 (BezPath::from_svg), vello/src/scene.rs:347-440 (Stroke).
// Restated from kurbo's published algorithms; parity for this layer is pinned only through the
// filled_circle / filled_square smoke goldens.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace kurbo {

struct Vec2 {
    double x = 0, y = 0;
};
struct Point {
    double x = 0, y = 0;
    Point() = default;
    Point(double x_, double y_) : x(x_), y(y_) {}
};
inline Point operator+(Point a, Vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline Point operator-(Point a, Vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline Vec2 operator-(Point a, Point b) { return {a.x - b.x, a.y - b.y}; }
inline Vec2 operator*(double s, Vec2 v) { return {s * v.x, s * v.y}; }
inline bool operator==(Point a, Point b) { return a.x == b.x && a.y == b.y; }

// Affine: [a b c d e f] maps (x,y) -> (a x + c y + e, b x + d y + f)
struct Affine {
    double c[6] = {1, 0, 0, 1, 0, 0};
    static Affine identity() { return Affine{}; }
    static Affine make(double a, double b, double cc, double d, double e, double f) {
        Affine r;
        r.c[0] = a; r.c[1] = b; r.c[2] = cc; r.c[3] = d; r.c[4] = e; r.c[5] = f;
        return r;
    }
    static Affine translate(double x, double y) { return make(1, 0, 0, 1, x, y); }
    static Affine scale(double s) { return make(s, 0, 0, s, 0, 0); }
    static Affine scale_non_uniform(double sx, double sy) { return make(sx, 0, 0, sy, 0, 0); }
    static Affine rotate(double th) {
        double s = std::sin(th), co = std::cos(th);
        return make(co, s, -s, co, 0, 0);
    }
    Affine operator*(const Affine &o) const {
        return make(c[0] * o.c[0] + c[2] * o.c[1], c[1] * o.c[0] + c[3] * o.c[1], c[0] * o.c[2] + c[2] * o.c[3],
                    c[1] * o.c[2] + c[3] * o.c[3], c[0] * o.c[4] + c[2] * o.c[5] + c[4],
                    c[1] * o.c[4] + c[3] * o.c[5] + c[5]);
    }
    Point operator*(Point p) const { return {c[0] * p.x + c[2] * p.y + c[4], c[1] * p.x + c[3] * p.y + c[5]}; }
    Affine pre_translate(double x, double y) const { return *this * translate(x, y); }
};

enum class Verb : uint8_t { MoveTo = 0, LineTo = 1, QuadTo = 2, CurveTo = 3, ClosePath = 4 };

struct PathEl {
    Verb verb;
    Point p[3];
};

struct BezPath {
    std::vector<PathEl> els;
    void move_to(Point p) { els.push_back({Verb::MoveTo, {p, {}, {}}}); }
    void line_to(Point p) { els.push_back({Verb::LineTo, {p, {}, {}}}); }
    void quad_to(Point p1, Point p2) { els.push_back({Verb::QuadTo, {p1, p2, {}}}); }
    void curve_to(Point p1, Point p2, Point p3) { els.push_back({Verb::CurveTo, {p1, p2, p3}}); }
    void close_path() { els.push_back({Verb::ClosePath, {{}, {}, {}}}); }
    void apply_affine(const Affine &a) {
        for (auto &e : els)
            for (auto &q : e.p) q = a * q;
    }
    // Parses SVG path data (kurbo svg.rs BezPath::from_svg).  Returns false on a syntax error.
    static bool from_svg(const std::string &d, BezPath &out);
};

struct Rect {
    double x0, y0, x1, y1;
    static Rect from_center_size(Point c, double w, double h) { return {c.x - 0.5 * w, c.y - 0.5 * h, c.x + 0.5 * w, c.y + 0.5 * h}; }
};
struct Circle {
    Point center;
    double radius;
};
struct Line {
    Point p0, p1;
};
struct RoundedRect {
    Rect rect;
    double radius;
};
struct Arc {
    Point center;
    Vec2 radii;
    double start_angle, sweep_angle, x_rotation;
    void append_iter(double tolerance, BezPath &out) const;
};

// Shape::path_elements(tolerance)
BezPath path_elements(const Rect &r, double tolerance);
BezPath path_elements(const Circle &c, double tolerance);
BezPath path_elements(const Line &l, double tolerance);
BezPath path_elements(const RoundedRect &r, double tolerance);

enum class Join : uint8_t { Bevel = 0, Miter = 1, Round = 2 };
enum class Cap : uint8_t { Butt = 0, Square = 1, Round = 2 };

// kurbo::Stroke; Stroke::new(width) defaults (kurbo 0.13): round join, round caps, miter limit 4.
struct Stroke {
    double width = 1.0;
    Join join = Join::Round;
    double miter_limit = 4.0;
    Cap start_cap = Cap::Round;
    Cap end_cap = Cap::Round;
    std::vector<double> dash_pattern;
    double dash_offset = 0.0;
    static Stroke make(double w) {
        Stroke s;
        s.width = w;
        return s;
    }
};

}  // namespace kurbo
