// printer.cpp - text forms of scalars, tensors and models; this defines what "same output" means
// for a .4th run.  Follows src/io/aio.cpp:38-57, aio_tensor.cpp:16-58,141-226, aio_model.cpp:65-141.
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

} // namespace t4
