// printer.cpp - text forms of scalars, tensors and models; this defines what "same output" means
// for a .4th run.  Follows src/io/aio.cpp:38-57, aio_tensor.cpp:16-58,141-226, aio_model.cpp:65-141.
#include <math.h>
#include "t4.h"
#include <iomanip>
#include <sstream>

namespace t4 {

static const int EDGE = 3, PREC = 4;                     // src/io/aio.h:80-82
static int THRES = 10;                                   // _thres: 10 for the printer, 1024 while a tensor is saved as text (aio_tensor.cpp:232-235)

std::string fmt_scalar(DU v, int base) {
    char buf[40];
    DU t, f = modff(v, &t);
    const bool dec = base == 10;
    if (dec && fabsf(f) > DU_EPS) { snprintf(buf, sizeof(buf), "%0.6g", v); return buf; }
    uint32_t n = dec ? (uint32_t)fabsf(v) : (uint32_t)v;
    int i = 33; buf[i] = '\0';
    do { uint8_t d = (uint8_t)(n % base); n /= base; buf[--i] = d > 9 ? (d - 10) + 'a' : d + '0'; } while (n && i);
    if (dec && v < 0) buf[--i] = '-';
    return &buf[i];
}
static std::string shape_s(Obj &o) {
    std::ostringstream s;
    s << '[';
    if (o.type == T_MODEL) s << ((int)((Model &)o).layer.size() - 1);
    else {
        Tensor &t = (Tensor &)o;
        switch (t.rank) {
        case 1: s << t.numel; break;
        case 2: s << t.H() << ',' << t.W(); break;
        case 3: s << "na"; break;
        default: s << t.N() << ',' << t.H() << ',' << t.W() << ',' << t.C(); break;
        }
    }
    s << ']';
    return s.str();
}
std::string fmt_objname(Obj &o, bool view) {
    static const char tn[2][4] = {{'T', 'N', 'D', 'X'}, {'t', 'n', 'd', 'x'}};
    std::ostringstream s;
    s << tn[view ? 1 : 0][o.type];
    if (o.type != T_MODEL) s << ((Tensor &)o).rank;
    s << shape_s(o);
    return s.str();
}
static std::string vec_s(const float *vd, uint32_t W, uint32_t C) {
    std::ostringstream o;
    o.flags(std::ios::showpos | std::ios::right | std::ios::fixed);
    o.precision(PREC);
    auto num = [&](const float *dx) { for (uint32_t k = 0; k < C; k++) o << (k > 0 ? "_" : " ") << *dx++; };
    o << "{";
    const uint32_t rw = (W <= (uint32_t)THRES) ? W : (W < (uint32_t)EDGE ? W : EDGE);
    for (uint32_t j = 0; j < rw; j++) num(vd + (size_t)j * C);
    const uint32_t x = W - rw;
    if (x > rw) o << " ...";
    for (uint32_t j = (x > rw ? x : rw); j < W; j++) num(vd + (size_t)j * C);
    o << " }";
    return o.str();
}
static std::string mat_s(const float *td, const uint32_t *shape) {
    const uint32_t H = shape[0], W = shape[1], C = shape[2];
    const uint64_t WC = (uint64_t)W * C;
    const uint32_t rh = H < (uint32_t)EDGE ? H : EDGE;
    std::ostringstream o;
    auto row = [&](uint32_t y1, const float *d) { o << vec_s(d, W, C) << (y1 == H ? "" : "\n\t"); };
    const float *d = td;
    for (uint32_t y = 0, y1 = 1; y < rh; y++, y1++, d += WC) row(y1, d);
    uint32_t ym = (H <= (uint32_t)THRES) ? rh : H - rh;
    if (ym > rh) o << "...\n\t"; else ym = rh;
    d = td + ym * WC;
    for (uint32_t y = ym, y1 = y + 1; y < H; y++, y1++, d += WC) row(y1, d);
    return o.str();
}
std::string fmt_tensor(Tensor &t, int thres) {
    std::vector<float> h; t.to_host(h);
    struct Thres { int old; explicit Thres(int v) : old(THRES) { if (v > 0) THRES = v; } ~Thres() { THRES = old; } } scope(thres);
    std::ostringstream o;
    switch (t.rank) {
    case 1: o << "vector" << shape_s(t) << " = " << vec_s(h.data(), (uint32_t)t.numel, 1); break;
    case 2: o << "matrix" << shape_s(t) << " = {\n\t" << mat_s(h.data(), t.shape) << " }"; break;
    case 4: {
        const int N = t.N();
        o << "tensor" << shape_s(t) << " = { {\n\t";
        const float *td = h.data();
        for (int n = 0; n < N; n++, td += t.HWC()) { o << mat_s(td, t.shape); o << ((n + 1) < N ? " } {\n\t" : ""); }
        o << " } }";
    } break;
    default: o << "tensor rank=" << t.rank << " not supported";
    }
    o << '\n';
    return o.str();
}
static std::string parm_s(Tensor &in, Tensor &out) {     // aio_model.cpp:103-141
    const int fn = in.grad_fn, S = in.stride[0];
    const DU p = in.xparm;
    std::ostringstream o;
    switch (fn) {
    case T4K_L_CONV: case T4K_L_DCONV: o << "bias=" << p << ", C=" << out.C() << ", K=" << in.grad[0]->H() << ", S=" << S << ", P=" << in.stride[2]; break;
    case T4K_L_LINEAR: o << "bias=" << p << ", H=" << in.grad[0]->H(); break;
    case T4K_L_SELU: case T4K_L_LEAKYRL: case T4K_L_ELU: o << "bias=" << p; break;
    case T4K_L_DROPOUT: o << "rate=" << p * 100.0 << '%'; break;
    case T4K_L_AVGPOOL: case T4K_L_MAXPOOL: case T4K_L_MINPOOL: o << S << "x" << S; break;
    case T4K_L_BATCHNM: o << "mtum=" << p; break;
    case T4K_L_USAMPLE: { const char *nm[] = {"nearest", "linear", "bilinear", "cubic"}; o << S << "x" << S << " " << nm[in.iparm & 3]; } break;
    default: break;
    }
    return o.str();
}
std::string fmt_parm(Tensor &in, Tensor &out) { return parm_s(in, out); }
std::string fmt_model(Model &m) {                        // aio_model.cpp:65-99
    std::ostringstream o;
    const int n = (int)m.layer.size();
    o << "NN Model[" << (n - 1) << "/128]\n";
    for (int i = 0; i < n; i++) {
        Tensor &in = m.at(i), &out = m.at(i + 1 < n ? i + 1 : i);
        o << '[' << std::setw(3) << i << "] " << LAYER_NAME[in.grad_fn] << ": " << fmt_objname(in, false);
        int sz = 0;
        for (int k = 0; k < 5; k++) sz += in.grad[k] ? (int)in.grad[k]->numel : 0;
        o << " #p=" << sz << ' ';
        for (int k = 0; k < 2 && in.grad[k]; k++) o << fmt_objname(*in.grad[k], false) << ' ';
        if (in.grad[4]) o << fmt_objname(*in.grad[4], false) << ' ';
        o << parm_s(in, out) << '\n';
    }
    return o.str();
}

// ---- the tensor debugger of the trace levels (`1 trace`: input preview and the loss derivative; `2 trace`: every layer tensor) - Tensor::show / _dump / _view,
// src/mu/tensor.cu:587-684.  A page [H][W][C] is printed channel by channel: numbers "%5.2f" with row sums, then (pages of more than 36 cells) a 16-shade
// character picture, two characters per cell (the left neighbour, and the mean of the two), scaled so that +-2 sigma of the whole tensor spans the shades.
static void dump_page(std::string &o, const float *v, uint32_t H, uint32_t W, uint32_t C) {
    char b[64];
    const float hw = (float)H * W, sr = sqrtf(hw);
    const uint32_t sh = (uint32_t)(hw / sr) + ((hw - sr * sr) > 0.f ? 1 : 0);
    const uint32_t h = W > 1 ? H : (hw < 36.0f ? 1 : sh), w = W > 1 ? W : (hw < 36.0f ? H : (uint32_t)sr);
    std::vector<float> csum(C, 0.f);
    for (uint32_t i = 0; i < h; i++) {
        o += "\n";
        float sum = 0.f;
        for (uint32_t k = 0; k < C; k++) {
            for (uint32_t j = 0; j < w; j++) {
                const uint64_t n = j + (uint64_t)i * w;
                if ((float)n >= hw) { o += " ...."; continue; }
                const float r = v[k + n * C];
                snprintf(b, sizeof(b), "%5.2f", r); o += b;
                sum += r; csum[k] += r;
            }
            o += "|";
        }
        snprintf(b, sizeof(b), "\xCE\xA3=%6.3f", sum); o += b;
    }
    if (h > 1) {
        o += "\n\xCE\xA3\xCE\xA3=";
        for (uint32_t k = 0; k < C; k++) { snprintf(b, sizeof(b), "%6.3f ", csum[k]); o += b; }
    }
}
static void view_page(std::string &o, const float *v, uint32_t H, uint32_t W, uint32_t C, float mean, float scale) {
    static const char shades[] = " `.-:;!+*ixekO#@";
    auto shade = [](float x) { const int i = (int)((x + 1.0) * 16 / 2); return shades[i < 0 ? 0 : (i < 16 ? i : 15)]; };
    char b[64];
    const uint64_t hw = (uint64_t)H * W, sr = (uint64_t)sqrtf((float)hw);
    const uint32_t sh = (uint32_t)(hw / sr) + ((hw - sr * sr) > 0 ? 1 : 0);
    const uint32_t w = W > 1 ? W : (hw < 36 ? H : (uint32_t)sr), h = W > 1 ? H : (hw < 36 ? 1 : sh);
    std::vector<float> csum(C, 0.f);
    for (uint32_t i = 0; i < h; i++) {
        o += "\n";
        for (uint32_t k = 0; k < C; k++) {
            for (uint32_t j = 0; j < w; j++) {
                const uint64_t n = j + (uint64_t)i * w;
                if (n >= hw) { o += "  "; continue; }
                const float r0 = v[k + (j > 0 ? n - 1 : n) * C], r1 = v[k + n * C];
                const float x0 = (r0 - mean) * scale, x1 = (float)(((r0 + r1) * 0.5) - mean) * scale;
                o += shade(x0); o += shade(x1);
                csum[k] += r1;
            }
            o += "|";
        }
    }
    if (h > 1) {
        o += "\n\xCE\xA3\xCE\xA3=";
        for (uint32_t k = 0; k < C; k++) { snprintf(b, sizeof(b), "%6.3f ", csum[k]); o += b; }
    }
    o += "\n";
}
std::string fmt_dump(Tensor &t) {                           // Tensor::_dump(data, H, W, C) of the whole tensor's first page (gradient.cu:70-74 hands it g.H(), g.W(), g.C())
    std::string o; std::vector<float> h; t.to_host(h);
    dump_page(o, h.data(), t.H(), t.W(), t.C());
    return o;
}
std::string fmt_show(Tensor &t, bool dump) {
    std::string o; char b[32];
    const uint32_t N = t.N(), H = t.H(), W = t.W(), C = t.C();
    const uint64_t hw = (uint64_t)H * W, page = hw * C;
    const float mean = t.avg(), scale = (float)(0.5 / t.std());      // (0.5 / std: P = 95 %)
    std::vector<float> h; t.to_host(h);
    for (uint32_t n = 0; n < N; n++) {
        const float *d = h.data() + (uint64_t)n * page;
        if (dump || hw < 100) { snprintf(b, sizeof(b), "\nn=%d", (int)n); o += b; dump_page(o, d, H, W, C); }
        if (hw > 36) view_page(o, d, H, W, C, mean, scale);
    }
    o += "\n";
    return o;
}

} // namespace t4
