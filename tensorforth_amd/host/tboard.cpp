// tboard.cpp - TensorBoard sink behind the words .tbinit .tbstep .scalar .histo .text .image .tile.
//
// Replaces the reference's src/tb (Summary / EventWriter, summary.cpp:18-118, writer.h:22-212) and the dispatch in
// src/sys.cpp:230-273.  The file format is TensorBoard's own (public): a sequence of records
//     uint64 length | uint32 masked_crc32c(length) | bytes[length] | uint32 masked_crc32c(bytes)
// each holding a serialized `Event` protobuf { 1: wall_time f64, 2: step i64, 3: file_version str | 5: Summary { 1: Value* } },
// Value { 1: tag, 2: simple_value f32 | 5: HistogramProto | 8: TensorProto (text, images), 9: SummaryMetadata }.
// Written from scratch as one flat byte builder; what follows the reference is behaviour, not code: the log layout
// <logdir>/<run_id>/events.out.tfevents.<time>.<host>.<pid>.0, the `brain.Event:2` header record, scalars as simple_value,
// and the histogram bucketing of writer.h:178-208 (an empty underflow bin at min, n equal-width bins, last limit = max + 1e-10),
// so the dashboards of t4_40a / t4_40b / t4_42a look the same.  .graph writes the model as a GraphDef event (Summary::graph summary.cpp:115-160,
// graph.h, writer.h:131-150), .embed the projector files of tb/projector.h (tensor / metadata TSV + projector_config.pbtxt).
//
// The sink is off unless a log directory is configured (`ten4 -t <logdir> [-r <run_id>]` or T4_TB_LOGDIR / T4_TB_RUN): the words then
// print the reference's "check TensorBoard param" message, exactly as the reference does without -t.
#include "t4.h"
#include <stdint.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <string>
#include <vector>

namespace t4 {

namespace {

typedef std::vector<uint8_t> Bytes;

// ---- crc32c (Castagnoli, reflected 0x82F63B78) and TensorBoard's mask
uint32_t crc32c(const uint8_t *d, size_t n) {
    static uint32_t tab[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0x82F63B78u ^ (c >> 1) : c >> 1; tab[i] = c; } init = true; }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = tab[(c ^ d[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
uint32_t masked(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// ---- protobuf wire format: just the four field kinds the event schema uses
struct PB {
    Bytes b;
    void varint(uint64_t v) { while (v >= 0x80) { b.push_back((uint8_t)(v | 0x80)); v >>= 7; } b.push_back((uint8_t)v); }
    void key(int field, int wire) { varint(((uint64_t)field << 3) | (uint64_t)wire); }
    void i64(int f, int64_t v) { key(f, 0); varint((uint64_t)v); }
    void f64(int f, double v) { key(f, 1); uint8_t t[8]; memcpy(t, &v, 8); b.insert(b.end(), t, t + 8); }
    void f32(int f, float v) { key(f, 5); uint8_t t[4]; memcpy(t, &v, 4); b.insert(b.end(), t, t + 4); }
    void bytes(int f, const uint8_t *d, size_t n) { key(f, 2); varint(n); b.insert(b.end(), d, d + n); }
    void str(int f, const std::string &s) { bytes(f, (const uint8_t *)s.data(), s.size()); }
    void msg(int f, const PB &m) { bytes(f, m.b.data(), m.b.size()); }
    void packed_f64(int f, const std::vector<double> &v) { key(f, 2); varint(v.size() * 8); for (double x : v) { uint8_t t[8]; memcpy(t, &x, 8); b.insert(b.end(), t, t + 8); } }
};

// ---- PNG with stored (uncompressed) deflate blocks: enough for the image dashboards, no zlib dependency
uint32_t crc32_ieee(const uint8_t *d, size_t n, uint32_t c = 0) {
    static uint32_t tab[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t x = i; for (int k = 0; k < 8; k++) x = (x & 1) ? 0xEDB88320u ^ (x >> 1) : x >> 1; tab[i] = x; } init = true; }
    c = ~c; for (size_t i = 0; i < n; i++) c = tab[(c ^ d[i]) & 0xFF] ^ (c >> 8); return ~c;
}
void be32(Bytes &o, uint32_t v) { o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v); }
void chunk(Bytes &o, const char *type, const Bytes &data) {
    be32(o, (uint32_t)data.size());
    Bytes t(type, type + 4); t.insert(t.end(), data.begin(), data.end());
    o.insert(o.end(), t.begin(), t.end()); be32(o, crc32_ieee(t.data(), t.size()));
}
Bytes png_rgb(int w, int h, const uint8_t *rgb) {
    Bytes raw; raw.reserve((size_t)h * (w * 3 + 1));
    for (int y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3); }
    Bytes z = { 0x78, 0x01 };
    uint32_t a = 1, b2 = 0;
    for (uint8_t v : raw) { a = (a + v) % 65521; b2 = (b2 + a) % 65521; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min((size_t)65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0); z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        if (raw.empty()) break;
    }
    be32(z, (b2 << 16) | a);
    Bytes o = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    Bytes hd; be32(hd, (uint32_t)w); be32(hd, (uint32_t)h); hd.push_back(8); hd.push_back(2); hd.push_back(0); hd.push_back(0); hd.push_back(0);
    chunk(o, "IHDR", hd); chunk(o, "IDAT", z); chunk(o, "IEND", Bytes());
    return o;
}

struct Sink {
    std::string root, run;
    std::vector<std::string> embeds;                     // embeddings of the current run (projector_config.pbtxt lists all of them)
    FILE *f = nullptr;
    int step = 0;
    double now() const { const char *t = getenv("T4_TB_FIXED_TIME"); return t ? atof(t) : (double)time(nullptr); }   // fixed clock: reproducible files for tests
    void record(const Bytes &ev) {
        if (!f) return;
        const uint64_t len = ev.size();
        const uint32_t lc = masked(crc32c((const uint8_t *)&len, 8)), dc = masked(crc32c(ev.data(), ev.size()));
        fwrite(&len, 8, 1, f); fwrite(&lc, 4, 1, f); fwrite(ev.data(), 1, ev.size(), f); fwrite(&dc, 4, 1, f);
        fflush(f);
    }
    bool open(const std::string &run_id) {                // Summary::init summary.cpp:18-28
        if (f) { fclose(f); f = nullptr; }
        run = run_id; embeds.clear();
        std::string dir_run = run; for (char &c : dir_run) if (c == ' ' || c == '/' || c == '\\') c = '_';
        mkdir(root.c_str(), 0755);
        const std::string dir = root + "/" + dir_run;
        mkdir(dir.c_str(), 0755);
        char host[256] = "localhost"; gethostname(host, sizeof(host)); host[255] = 0;
        for (char *q = host; *q; q++) if (*q == '/' || *q == '\\' || *q == ':') *q = '_';
        char name[1024]; snprintf(name, sizeof(name), "%s/events.out.tfevents.%ld.%s.%d.0", dir.c_str(), (long)now(), host, (int)getpid());
        f = fopen(name, "wb");
        if (!f) { hprintf("  tb#init cannot open %s\n", name); return false; }
        PB ev; ev.f64(1, now()); ev.i64(2, 0); ev.str(3, "brain.Event:2");
        record(ev.b);
        return true;
    }
    void value(const PB &val) {                          // Event { wall_time, step, summary { value } }
        PB sum; sum.msg(1, val);
        PB ev; ev.f64(1, now()); ev.i64(2, step); ev.msg(5, sum);
        record(ev.b);
    }
    // SummaryMetadata { 1: PluginData { 1: plugin_name }, 4: data_class } (schema.h:72-119; the histogram's carries no data_class)
    static PB meta(const char *plugin, int data_class = 0) { PB pd; pd.str(1, plugin); PB md; md.msg(1, pd); if (data_class) md.i64(4, data_class); return md; }
};
Sink *g_tb = nullptr;

} // namespace

bool tb_configure(const char *logdir, const char *run_id) {   // `ten4 -t <logdir> -r <run_id>` (reference: src/ten4.cu options -t / -r)
    if (!logdir || !*logdir) return false;
    if (!g_tb) g_tb = new Sink();
    g_tb->root = logdir;
    return g_tb->open(run_id && *run_id ? run_id : "run1");
}
bool tb_active() {
    if (!g_tb) { const char *d = getenv("T4_TB_LOGDIR"); if (d && *d) tb_configure(d, getenv("T4_TB_RUN")); }
    return g_tb && g_tb->f;
}
void tb_init(const char *run_id) { if (tb_active()) g_tb->open(run_id); }                // .tbinit: a new run directory under the configured logdir
void tb_step(int i) { if (tb_active()) g_tb->step = i; }
void tb_scalar(const char *tag, float v) {               // EventWriter::add_scalar writer.h:50-56
    if (!tb_active()) return;
    PB val; val.str(1, tag); val.f32(2, v);
    g_tb->value(val);
}
void tb_text(const char *tag, const char *txt) {         // add_text writer.h:59-67 + schema.h:37-45,88-98: scalar DT_STRING tensor (no shape), plugin "text", DATA_CLASS_TENSOR
    if (!tb_active()) return;
    PB ten; ten.i64(1, 7 /* DT_STRING */); ten.str(8, txt);
    PB val; val.str(1, tag); val.msg(9, Sink::meta("text", 2)); val.msg(8, ten);
    g_tb->value(val);
}
void tb_histo(const char *tag, Tensor &t, int nb) {      // Summary::histo summary.cpp:103-112 + add_histo / _buckets writer.h:90-118,178-208
    if (!tb_active() || t.numel == 0) return;
    if (nb < 1) nb = 30;
    std::vector<float> h; t.to_host(h);
    double vsum = 0, vsq = 0, vmin = h[0], vmax = h[0];
    for (float x : h) { vsum += x; vsq += (double)x * x; vmin = std::min(vmin, (double)x); vmax = std::max(vmax, (double)x); }
    std::vector<double> limits, counts;
    if (vmin == vmax) { limits.push_back(vmin + 1e-10); counts.push_back((double)h.size()); }
    else {
        const double bw = (vmax - vmin) / nb;
        limits.push_back(vmin); counts.push_back(0.0);                      // empty underflow bin: the left edge is drawn
        for (int i = 0; i < nb; i++) { limits.push_back(vmin + (i + 1) * bw); counts.push_back(0.0); }
        limits.back() = vmax + 1e-10;
        for (float x : h) { const int b = std::max(0, std::min(nb - 1, (int)((x - vmin) / bw))); counts[b + 1] += 1.0; }
    }
    PB hp; hp.f64(1, vmin); hp.f64(2, vmax); hp.f64(3, (double)h.size()); hp.f64(4, vsum); hp.f64(5, vsq); hp.packed_f64(6, limits); hp.packed_f64(7, counts);
    PB val; val.str(1, tag); val.msg(9, Sink::meta("histograms")); val.msg(5, hp);
    g_tb->value(val);
}
static void tb_png(const char *tag, int w, int h, const std::vector<uint8_t> &rgb) {
    // add_image writer.h:69-80 + schema.h:47-68,100-110: the Time-Series form - DT_STRING tensor of shape [3] holding width, height (decimal
    // text) and the PNG bytes, then plugin "images" with DATA_CLASS_BLOB_SEQUENCE (tensor in front of the metadata, as the reference writes it)
    const Bytes png = png_rgb(w, h, rgb.data());
    PB dim; dim.i64(1, 3); PB shape; shape.msg(2, dim);
    PB ten; ten.i64(1, 7 /* DT_STRING */); ten.msg(2, shape); ten.str(8, std::to_string(w)); ten.str(8, std::to_string(h)); ten.bytes(8, png.data(), png.size());
    PB val; val.str(1, tag); val.msg(8, ten); val.msg(9, Sink::meta("images", 3));
    g_tb->value(val);
}
void tb_tile(const char *tag, Tensor &t, int per_row) {   // Summary::tile summary.cpp:66-101: N images on a grid, 2-pixel border, x 256 grey / RGB
    if (!tb_active() || t.numel == 0) return;
    if (per_row < 1) per_row = 1;
    const int B = 2, N = t.N(), H = t.H(), W = t.W(), C = t.C();
    const int WT = (W + B) * per_row + B, HT = (H + B) * ((N + per_row - 1) / per_row) + B;
    std::vector<float> hx; t.to_host(hx);
    std::vector<uint8_t> px((size_t)HT * WT * 3, 0);
    for (int n = 0; n < N; n++) {
        const float *v = &hx[(size_t)n * H * W * C];
        const int ty = n / per_row, tx = n % per_row;
        for (int y = 0; y < H; y++) {
            uint8_t *q = &px[((size_t)(ty * (H + B) + y + B) * WT + tx * (W + B) + B) * 3];
            for (int x = 0; x < W; x++, v += C)
                for (int c = 0; c < 3; c++) { const float vx = v[c < C ? c : C - 1] * 256.0f; *q++ = (uint8_t)std::min(255.0f, std::max(vx, 0.0f)); }
        }
    }
    tb_png(tag, WT, HT, px);
}
void tb_image(const char *tag, Tensor &t) {              // Summary::image summary.cpp:30-64: one image per sample, (x + mean) * 64 / std
    if (!tb_active() || t.numel == 0) return;
    const int N = t.N(), H = t.H(), W = t.W(), C = t.C();
    const float mean = t.avg(), sd = t.std(), scale = 64.0f / (sd > 0 ? sd : 1.0f);
    std::vector<float> hx; t.to_host(hx);
    std::vector<uint8_t> px((size_t)H * W * 3);
    for (int n = 0; n < N; n++) {
        const float *v = &hx[(size_t)n * H * W * C];
        for (int i = 0; i < H * W; i++, v += C)
            for (int c = 0; c < 3; c++) { const float vx = (v[c < C ? c : C - 1] + mean) * scale; px[(size_t)i * 3 + c] = (uint8_t)std::min(255.0f, std::max(vx, 0.0f)); }
        tb_png(tag, W, H, px);
    }
}
// .graph ( N -- ): Event { 2: step = 0, 4: GraphDef { 1: NodeDef* } } - no wall_time, as the reference writes it (writer.h:139-150).
// NodeDef { 1: name, 2: op, 3: input, 5: attr { 1: key, 2: AttrValue } }: a Placeholder "input" with the model's input shape, then one
// node per layer tensor named <Op>_<i>/<layer name> whose input is the node in front; attrs dtype = DT_FLOAT and shape = [N,H,W,C].
void tb_graph(Model &m) {
    if (!tb_active()) return;
    static const char *op[] = { "Output", "Conv2D", "MatMul", "Reshape", "Relu", "Tanh", "Sigmoid", "Selu", "LeakyRelu", "Elu", "Dropout", "Softmax", "LogSoftmax",
                                "AvgPool", "MaxPool", "MinPool", "BatchNorm", "UpSample", "UpSample" };     // summary.cpp:120-125 (+ dconv2d, which the reference's table lacks)
    auto attrs = [](PB &node, Tensor &t) {
        PB dt; dt.i64(6, 1);                                                       // AttrValue.type = DT_FLOAT
        PB a1; a1.str(1, "dtype"); a1.msg(2, dt); node.msg(5, a1);
        PB dims;
        for (uint32_t d : { t.N(), t.H(), t.W(), t.C() }) { PB sz; sz.i64(1, d); dims.msg(2, sz); }   // TensorShapeProto.dim { size }
        PB av; av.msg(7, dims);                                                    // AttrValue.shape
        PB a2; a2.str(1, "shape"); a2.msg(2, av); node.msg(5, a2);
    };
    auto name_of = [&](int i) { Tensor &t = m.at(i); return std::string(op[t.grad_fn]) + "_" + std::to_string(i) + "/" + LAYER_NAME[t.grad_fn]; };
    PB graph;
    { PB n0; n0.str(1, "input"); n0.str(2, "Placeholder"); attrs(n0, m.at(0)); graph.msg(1, n0); }
    const int n = (int)m.layer.size();
    for (int i = 0; i < n; i++) {
        PB nd; nd.str(1, name_of(i)); nd.str(2, op[m.at(i).grad_fn]); nd.str(3, i == 0 ? std::string("input") : name_of(i - 1));
        attrs(nd, m.at(i)); graph.msg(1, nd);
    }
    PB ev; ev.i64(2, 0); ev.msg(4, graph);
    g_tb->record(ev.b);
}
// .embed ( T tag -- ): <run dir>/<tag>_tensors.tsv (one row per sample, tab-separated, operator<< float format), <tag>_metadata.tsv
// (<tag>.<n> per row) and projector_config.pbtxt listing every embedding of the run so far (projector.h:31-71, summary.cpp:162-177)
void tb_embed(const char *tag_in, Tensor &t) {
    if (!tb_active() || t.numel == 0) return;
    auto esc = [](std::string v) { for (char &c : v) if (c == ' ' || c == '/' || c == '\\') c = '_'; return v; };
    const std::string dir = g_tb->root + "/" + esc(g_tb->run), tag = esc(tag_in), base = dir + "/" + tag;
    std::vector<float> h; t.to_host(h);
    const size_t N = t.N(), D = (size_t)t.HWC();
    FILE *f = fopen((base + "_tensors.tsv").c_str(), "w"); if (!f) { hprintf("  tb#embed cannot write %s_tensors.tsv\n", base.c_str()); return; }
    for (size_t n = 0; n < N; n++) { for (size_t i = 0; i < D; i++) fprintf(f, "%s%g", i ? "\t" : "", h[n * D + i]); fputc('\n', f); }   // %g == ostream << float
    fclose(f);
    f = fopen((base + "_metadata.tsv").c_str(), "w"); if (!f) return;
    for (size_t n = 0; n < N; n++) fprintf(f, "%s.%zu\n", tag.c_str(), n);
    fclose(f);
    g_tb->embeds.push_back(tag);
    f = fopen((dir + "/projector_config.pbtxt").c_str(), "w"); if (!f) return;
    for (const std::string &e : g_tb->embeds)
        fprintf(f, "embeddings {\n  tensor_name: \"%s\"\n  tensor_path: \"%s/%s_tensors.tsv\"\n  metadata_path: \"%s/%s_metadata.tsv\"\n}\n", e.c_str(), dir.c_str(), e.c_str(), dir.c_str(), e.c_str());
    fclose(f);
}
void tb_close() { if (g_tb && g_tb->f) { fclose(g_tb->f); g_tb->f = nullptr; } }

} // namespace t4
