// See kurbo.hpp.  Restated from kurbo's published algorithms (circle.rs, rect.rs, arc.rs, svg.rs).
#include "kurbo.hpp"

#include <cctype>
#include <cstdlib>

namespace kurbo {

static constexpr double PI = 3.14159265358979323846;
static constexpr double FRAC_PI_2 = PI / 2.0;
static constexpr double FRAC_PI_4 = PI / 4.0;

BezPath path_elements(const Rect &r, double) {
    BezPath p;
    p.move_to({r.x0, r.y0});
    p.line_to({r.x1, r.y0});
    p.line_to({r.x1, r.y1});
    p.line_to({r.x0, r.y1});
    p.close_path();
    return p;
}

BezPath path_elements(const Line &l, double) {
    BezPath p;
    p.move_to(l.p0);
    p.line_to(l.p1);
    return p;
}

// kurbo circle.rs: n = 4 arcs with arm 0.551915024494 while |r|/tolerance < 1/1.9608e-4,
// else n = ceil((1.1163 * scaled_err)^(1/6)), arm = 4/3 tan(pi/(4n)).
BezPath path_elements(const Circle &c, double tolerance) {
    BezPath p;
    double scaled_err = std::fabs(c.radius) / tolerance;
    size_t n;
    double arm_len;
    if (scaled_err < 1.0 / 1.9608e-4) {
        n = 4;
        arm_len = 0.551915024494;
    } else {
        n = (size_t)std::ceil(std::pow(1.1163 * scaled_err, 1.0 / 6.0));
        arm_len = (4.0 / 3.0) * std::tan(FRAC_PI_4 / (double)n);
    }
    double delta_th = 2.0 * PI / (double)n;
    double a = c.radius, x = c.center.x, y = c.center.y;
    p.move_to({x + a, y});
    for (size_t ix = 1; ix <= n; ix++) {
        double th1 = delta_th * (double)ix;
        double th0 = th1 - delta_th;
        double s0 = std::sin(th0), c0 = std::cos(th0);
        double s1, c1;
        if (ix == n) {
            s1 = 0.0;
            c1 = 1.0;
        } else {
            s1 = std::sin(th1);
            c1 = std::cos(th1);
        }
        p.curve_to({x + a * (c0 - arm_len * s0), y + a * (s0 + arm_len * c0)},
                   {x + a * (c1 + arm_len * s1), y + a * (s1 - arm_len * c1)}, {x + a * c1, y + a * s1});
    }
    p.close_path();
    return p;
}

static Vec2 rotate_pt(Vec2 pt, double angle) {
    double s = std::sin(angle), c = std::cos(angle);
    return {pt.x * c - pt.y * s, pt.x * s + pt.y * c};
}
static Vec2 sample_ellipse(Vec2 radii, double x_rotation, double angle) {
    double s = std::sin(angle), c = std::cos(angle);
    return rotate_pt({radii.x * c, radii.y * s}, x_rotation);
}

// kurbo arc.rs Arc::append_iter
void Arc::append_iter(double tolerance, BezPath &out) const {
    double sign = sweep_angle > 0 ? 1.0 : (sweep_angle < 0 ? -1.0 : 0.0);
    double scaled_err = std::fmax(radii.x, radii.y) / tolerance;
    double n_err = std::fmax(std::pow(1.1163 * scaled_err, 1.0 / 6.0), 3.999999);
    double nf = std::ceil(n_err * std::fabs(sweep_angle) * (1.0 / (2.0 * PI)));
    double angle_step = sweep_angle / nf;
    size_t n = (size_t)nf;
    double arm_len = (4.0 / 3.0) * std::tan(std::fabs(0.25 * angle_step)) * sign;
    double angle0 = start_angle;
    Vec2 p0 = sample_ellipse(radii, x_rotation, angle0);
    for (size_t i = 0; i < n; i++) {
        double angle1 = angle0 + angle_step;
        Vec2 t0 = sample_ellipse(radii, x_rotation, angle0 + FRAC_PI_2);
        Vec2 p1 = {p0.x + arm_len * t0.x, p0.y + arm_len * t0.y};
        Vec2 p3 = sample_ellipse(radii, x_rotation, angle1);
        Vec2 t1 = sample_ellipse(radii, x_rotation, angle1 + FRAC_PI_2);
        Vec2 p2 = {p3.x - arm_len * t1.x, p3.y - arm_len * t1.y};
        out.curve_to(center + p1, center + p2, center + p3);
        angle0 = angle1;
        p0 = p3;
    }
}

BezPath path_elements(const RoundedRect &rr, double tolerance) {
    BezPath p;
    const Rect &r = rr.rect;
    double w = std::fabs(r.x1 - r.x0), h = std::fabs(r.y1 - r.y0);
    double rad = std::fmin(std::fabs(rr.radius), 0.5 * std::fmin(w, h));
    p.move_to({r.x0, r.y0 + rad});
    auto corner = [&](int i, Point c) {
        Arc a{c, {rad, rad}, FRAC_PI_2 * (double)i, FRAC_PI_2, 0.0};
        a.append_iter(tolerance, p);
    };
    corner(2, {r.x0 + rad, r.y0 + rad});
    p.line_to({r.x1 - rad, r.y0});
    corner(3, {r.x1 - rad, r.y0 + rad});
    p.line_to({r.x1, r.y1 - rad});
    corner(0, {r.x1 - rad, r.y1 - rad});
    p.line_to({r.x0 + rad, r.y1});
    corner(1, {r.x0 + rad, r.y1 - rad});
    p.close_path();
    return p;
}

// ---- SVG path data (kurbo svg.rs) ----
namespace {
struct Lexer {
    const char *s;
    size_t n, i = 0;
    void skip_ws() {
        while (i < n && (std::isspace((unsigned char)s[i]) || s[i] == ',')) i++;
    }
    bool at_end() {
        skip_ws();
        return i >= n;
    }
    bool peek_is_number() {
        skip_ws();
        if (i >= n) return false;
        char c = s[i];
        return std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.';
    }
    bool number(double &out) {
        skip_ws();
        size_t st = i;
        if (i < n && (s[i] == '-' || s[i] == '+')) i++;
        bool digits = false;
        while (i < n && std::isdigit((unsigned char)s[i])) { i++; digits = true; }
        if (i < n && s[i] == '.') {
            i++;
            while (i < n && std::isdigit((unsigned char)s[i])) { i++; digits = true; }
        }
        if (!digits) return false;
        if (i < n && (s[i] == 'e' || s[i] == 'E')) {
            size_t save = i;
            i++;
            if (i < n && (s[i] == '-' || s[i] == '+')) i++;
            if (i < n && std::isdigit((unsigned char)s[i])) {
                while (i < n && std::isdigit((unsigned char)s[i])) i++;
            } else {
                i = save;
            }
        }
        out = std::strtod(std::string(s + st, i - st).c_str(), nullptr);
        return true;
    }
    bool flag(bool &out) {
        skip_ws();
        if (i >= n || (s[i] != '0' && s[i] != '1')) return false;
        out = s[i] == '1';
        i++;
        return true;
    }
};

// kurbo arc.rs Arc::from_svg_arc
bool svg_arc_to_arc(Point from, Point to, Vec2 radii, double x_rotation, bool large_arc, bool sweep, Arc &out) {
    if (std::fabs(radii.x) <= 1e-5 || std::fabs(radii.y) <= 1e-5 || from == to) return false;
    double rx = std::fabs(radii.x), ry = std::fabs(radii.y);
    double xr = std::fmod(x_rotation, 2.0 * PI);
    double sin_phi = std::sin(xr), cos_phi = std::cos(xr);
    double hd_x = (from.x - to.x) * 0.5, hd_y = (from.y - to.y) * 0.5;
    double hs_x = (from.x + to.x) * 0.5, hs_y = (from.y + to.y) * 0.5;
    double px = cos_phi * hd_x + sin_phi * hd_y, py = -sin_phi * hd_x + cos_phi * hd_y;
    double rf = px * px / (rx * rx) + py * py / (ry * ry);
    if (rf > 1.0) {
        double sc = std::sqrt(rf);
        rx *= sc;
        ry *= sc;
    }
    double rxry = rx * ry, rxpy = rx * py, rypx = ry * px;
    double sum_of_sq = rxpy * rxpy + rypx * rypx;
    if (sum_of_sq == 0.0) return false;
    double sign_coe = (large_arc == sweep) ? -1.0 : 1.0;
    double coe = sign_coe * std::sqrt(std::fabs((rxry * rxry - sum_of_sq) / sum_of_sq));
    double tcx = coe * rxpy / ry, tcy = -coe * rypx / rx;
    Point center{cos_phi * tcx - sin_phi * tcy + hs_x, sin_phi * tcx + cos_phi * tcy + hs_y};
    double sx = (px - tcx) / rx, sy = (py - tcy) / ry;
    double ex = (-px - tcx) / rx, ey = (-py - tcy) / ry;
    double start_angle = std::atan2(sy, sx);
    double sweep_angle = std::fmod(std::atan2(ey, ex) - start_angle, 2.0 * PI);
    if (sweep && sweep_angle < 0.0) sweep_angle += 2.0 * PI;
    else if (!sweep && sweep_angle > 0.0) sweep_angle -= 2.0 * PI;
    out = Arc{center, {rx, ry}, start_angle, sweep_angle, x_rotation};
    return true;
}
}  // namespace

bool BezPath::from_svg(const std::string &d, BezPath &path) {
    Lexer lx{d.c_str(), d.size()};
    Point last_pt{0, 0}, start_pt{0, 0}, last_ctrl{0, 0};
    bool have_ctrl_cubic = false, have_ctrl_quad = false;
    char cmd = 0;
    bool implicit_moveto = false;
    while (!lx.at_end()) {
        char c = lx.s[lx.i];
        if (std::isalpha((unsigned char)c)) {
            cmd = c;
            lx.i++;
            implicit_moveto = false;
        } else if (cmd == 0) {
            return false;
        } else if (implicit_moveto) {
            cmd = (cmd == 'M') ? 'L' : (cmd == 'm' ? 'l' : cmd);
        }
        bool rel = std::islower((unsigned char)cmd);
        auto get_pt = [&](Point &o) -> bool {
            double x, y;
            if (!lx.number(x) || !lx.number(y)) return false;
            o = rel ? Point{last_pt.x + x, last_pt.y + y} : Point{x, y};
            return true;
        };
        bool cubic_now = false, quad_now = false;
        switch (cmd) {
        case 'M': case 'm': {
            Point p;
            if (!get_pt(p)) return false;
            path.move_to(p);
            last_pt = start_pt = p;
            implicit_moveto = true;
            break;
        }
        case 'L': case 'l': {
            Point p;
            if (!get_pt(p)) return false;
            path.line_to(p);
            last_pt = p;
            break;
        }
        case 'H': case 'h': {
            double x;
            if (!lx.number(x)) return false;
            last_pt = {rel ? last_pt.x + x : x, last_pt.y};
            path.line_to(last_pt);
            break;
        }
        case 'V': case 'v': {
            double y;
            if (!lx.number(y)) return false;
            last_pt = {last_pt.x, rel ? last_pt.y + y : y};
            path.line_to(last_pt);
            break;
        }
        case 'C': case 'c': {
            Point p1, p2, p3;
            if (!get_pt(p1) || !get_pt(p2) || !get_pt(p3)) return false;
            path.curve_to(p1, p2, p3);
            last_ctrl = p2;
            last_pt = p3;
            cubic_now = true;
            break;
        }
        case 'S': case 's': {
            Point p2, p3;
            if (!get_pt(p2) || !get_pt(p3)) return false;
            Point p1 = have_ctrl_cubic ? Point{2 * last_pt.x - last_ctrl.x, 2 * last_pt.y - last_ctrl.y} : last_pt;
            path.curve_to(p1, p2, p3);
            last_ctrl = p2;
            last_pt = p3;
            cubic_now = true;
            break;
        }
        case 'Q': case 'q': {
            Point p1, p2;
            if (!get_pt(p1) || !get_pt(p2)) return false;
            path.quad_to(p1, p2);
            last_ctrl = p1;
            last_pt = p2;
            quad_now = true;
            break;
        }
        case 'T': case 't': {
            Point p2;
            if (!get_pt(p2)) return false;
            Point p1 = have_ctrl_quad ? Point{2 * last_pt.x - last_ctrl.x, 2 * last_pt.y - last_ctrl.y} : last_pt;
            path.quad_to(p1, p2);
            last_ctrl = p1;
            last_pt = p2;
            quad_now = true;
            break;
        }
        case 'A': case 'a': {
            double rx, ry, rot;
            bool large, sweep;
            Point p;
            if (!lx.number(rx) || !lx.number(ry) || !lx.number(rot) || !lx.flag(large) || !lx.flag(sweep) || !get_pt(p))
                return false;
            Arc arc;
            if (svg_arc_to_arc(last_pt, p, {rx, ry}, rot * PI / 180.0, large, sweep, arc)) {
                arc.append_iter(0.1, path);
            } else {
                path.line_to(p);
            }
            last_pt = p;
            break;
        }
        case 'Z': case 'z':
            path.close_path();
            last_pt = start_pt;
            break;
        default:
            return false;
        }
        have_ctrl_cubic = cubic_now;
        have_ctrl_quad = quad_now;
        if (cmd == 'Z' || cmd == 'z') {
            // a number directly after z is a syntax error; next token must be a command
            if (lx.peek_is_number()) return false;
        }
    }
    return true;
}

}  // namespace kurbo
