"""TensorBoard sink (words .tbinit .tbstep .scalar .histo .text .tile .graph .embed, host/tboard.cpp).

CPU test through the oracle-backed VM (same host sources as the product): a script logs a scalar, a histogram, a text and an image
tile; the tfevents file is then (1) checked record by record against TensorBoard's framing - length, masked crc32c of the length,
payload, masked crc32c of the payload - with a crc32c written here, and (2) compared BYTE FOR BYTE with the file this test builds
itself from the protobuf schema (Event / Summary / HistogramProto / TensorProto / Image) with its own encoder and the bucket rule of
the reference's writer (underflow bin at min, n equal bins, last limit max + 1e-10).  The clock is pinned with T4_TB_FIXED_TIME."""
import glob
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from vm_util import ROOT, TEN4, TEN4_ORACLE

T0 = 1700000000.0
SCRIPT = '''0 trace
3 .tbstep
0.5 s" train/loss" .scalar
8 vector{ -1 0 0.25 0.5 1 2 2 3 } 4 s" nn/w" .histo
: note s" epoch three" s" notes" .text ;
note
7 .tbstep
2 2 3 1 tensor ={ 0 0.25 0.5 0.75 1 0.125 0.5 0.5 0.5 0.5 0.5 0.5 } 2 s" imgs" .tile
1.5 s" train/loss" .scalar
bye
'''


def crc32c(data):
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (0x82F63B78 ^ (c >> 1)) if c & 1 else c >> 1
        tab.append(c)
    c = 0xFFFFFFFF
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ---- a second, independent protobuf encoder (the test's statement of the schema)
def varint(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80]); v >>= 7
    return out + bytes([v])


def key(f, w): return varint((f << 3) | w)
def f_i64(f, v): return key(f, 0) + varint(v)
def f_f64(f, v): return key(f, 1) + struct.pack("<d", v)
def f_f32(f, v): return key(f, 5) + struct.pack("<f", v)
def f_len(f, b): return key(f, 2) + varint(len(b)) + b
def f_pk64(f, vs): return f_len(f, b"".join(struct.pack("<d", x) for x in vs))


def event(step, value):
    return f_f64(1, T0) + f_i64(2, step) + f_len(5, f_len(1, value))


def meta(plugin, data_class=0):                            # SummaryMetadata { 1: PluginData { 1: plugin_name }, 4: data_class } (reference src/tb/schema.h:72-119)
    return f_len(1, f_len(1, plugin.encode())) + (f_i64(4, data_class) if data_class else b"")


def histo_value(tag, xs, nb):
    xs = np.asarray(xs, np.float32).astype(np.float64)
    vmin, vmax = xs.min(), xs.max()
    bw = (vmax - vmin) / nb
    limits = [vmin] + [vmin + (i + 1) * bw for i in range(nb)]; limits[-1] = vmax + 1e-10
    counts = [0.0] * (nb + 1)
    for x in xs:
        counts[max(0, min(nb - 1, int((x - vmin) / bw))) + 1] += 1.0
    hp = f_f64(1, vmin) + f_f64(2, vmax) + f_f64(3, float(len(xs))) + f_f64(4, xs.sum()) + f_f64(5, (xs * xs).sum()) + f_pk64(6, limits) + f_pk64(7, counts)
    return f_len(1, tag.encode()) + f_len(9, meta("histograms")) + f_len(5, hp)


def png_stored(w, h, rgb):
    raw = b"".join(b"\x00" + rgb[y * w * 3:(y + 1) * w * 3] for y in range(h))
    z = b"\x78\x01" + b"\x01" + struct.pack("<HH", len(raw), ~len(raw) & 0xFFFF) + raw + struct.pack(">I", zlib.adler32(raw))
    def ch(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    return b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + ch(b"IDAT", z) + ch(b"IEND", b"")


def tile_pixels(t, per_row):
    n, h, w, c = t.shape; B = 2
    WT, HT = (w + B) * per_row + B, (h + B) * ((n + per_row - 1) // per_row) + B
    px = np.zeros((HT, WT, 3), np.uint8)
    for i in range(n):
        ty, tx = divmod(i, per_row)
        v = np.clip(t[i] * np.float32(256.0), 0, 255).astype(np.uint8)          # [h,w,c]
        for ch_ in range(3):
            px[ty * (h + B) + B: ty * (h + B) + B + h, tx * (w + B) + B: tx * (w + B) + B + w, ch_] = v[:, :, min(ch_, c - 1)]
    return WT, HT, px.tobytes()


def records(blob):
    out = []; off = 0
    while off < len(blob):
        (n,) = struct.unpack_from("<Q", blob, off)
        (lc,) = struct.unpack_from("<I", blob, off + 8)
        data = blob[off + 12: off + 12 + n]
        (dc,) = struct.unpack_from("<I", blob, off + 12 + n)
        assert lc == masked(crc32c(blob[off:off + 8])), "length crc"
        assert dc == masked(crc32c(data)), "payload crc"
        out.append(data); off += 16 + n
    return out


@pytest.fixture(scope="module")
def oracle_vm():
    if not os.path.exists(TEN4_ORACLE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ten4_oracle"], check=True, capture_output=True)
    return TEN4_ORACLE


def test_crc32c_known_answer():
    assert crc32c(b"123456789") == 0xE3069283                  # the standard check value of CRC-32C (Castagnoli)


def test_events_file_is_byte_exact(oracle_vm, tmp_path):
    _byte_exact(oracle_vm, tmp_path)


@pytest.mark.gpu
def test_events_file_is_byte_exact_from_the_product_vm(tmp_path):
    _byte_exact(TEN4, tmp_path)                                 # tensors come back from HBM here: same bytes


def _byte_exact(binary, tmp_path):
    env = dict(os.environ, T4_SEED="1", T4_TB_FIXED_TIME=str(int(T0)))
    r = subprocess.run([binary, "-t", str(tmp_path), "-r", "run A/1"], input=SCRIPT, capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "check TensorBoard param" not in r.stdout, r.stdout
    files = glob.glob(os.path.join(str(tmp_path), "run_A_1", "events.out.tfevents.%d.*.0" % int(T0)))   # run id escaped as the reference does
    assert len(files) == 1, os.listdir(str(tmp_path))
    blob = open(files[0], "rb").read()
    recs = records(blob)                                        # framing + both crcs of every record
    tile = np.array([0, 0.25, 0.5, 0.75, 1, 0.125, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5], np.float32).reshape(2, 2, 3, 1)
    WT, HT, px = tile_pixels(tile, 2)
    # the reference's writer as of this tree (src/tb/writer.h:59-80, schema.h:37-68): text = scalar DT_STRING tensor + plugin "text" with DATA_CLASS_TENSOR;
    # images = DT_STRING tensor of shape [3] { width, height, PNG } in front of plugin "images" with DATA_CLASS_BLOB_SEQUENCE.  tests/test_refhost_parity.py
    # holds the same bytes as written by the reference's own code.
    img_tensor = f_i64(1, 7) + f_len(2, f_len(2, f_i64(1, 3))) + f_len(8, str(WT).encode()) + f_len(8, str(HT).encode()) + f_len(8, png_stored(WT, HT, px))
    text_tensor = f_i64(1, 7) + f_len(8, b"epoch three")
    want = [
        f_f64(1, T0) + f_i64(2, 0) + f_len(3, b"brain.Event:2"),
        event(3, f_len(1, b"train/loss") + f_f32(2, 0.5)),
        event(3, histo_value("nn/w", [-1, 0, 0.25, 0.5, 1, 2, 2, 3], 4)),
        event(3, f_len(1, b"notes") + f_len(9, meta("text", 2)) + f_len(8, text_tensor)),
        event(7, f_len(1, b"imgs") + f_len(8, img_tensor) + f_len(9, meta("images", 3))),
        event(7, f_len(1, b"train/loss") + f_f32(2, 1.5)),
    ]
    assert len(recs) == len(want)
    for i, (g, w) in enumerate(zip(recs, want)):
        assert g == w, "record %d differs:\n got %s\nwant %s" % (i, g.hex(), w.hex())
    # and the image decodes with a real inflater to the grid the tile word lays out
    png = png_stored(WT, HT, px)
    idat = png[png.index(b"IDAT") + 4: png.index(b"IEND") - 8]
    raw = zlib.decompress(idat)
    assert len(raw) == HT * (WT * 3 + 1)


GRAPH_SCRIPT = '''0 trace
2 8 8 1 nn.model 0.5 3 conv2d 2 maxpool relu flatten 5 linear softmax constant net
net .graph
3 2 2 1 tensor ={ 0.5 -1 2 0.25 1 1 1 1 0 0 0 0.0000001 } s" act/l 1" .embed
2 3 matrix{ 1 2 3 4 5 6 } s" w" .embed
bye
'''


def _graph_and_embed(binary, tmp_path):
    """.graph / .embed (tenvm.cpp:610-611, Summary::graph / embed summary.cpp:115-177): the GraphDef event rebuilt here from the
    protobuf schema (NodeDef 1 name, 2 op, 3 input, 5 attr{key, AttrValue}; AttrValue 6 type / 7 shape; TensorShapeProto 2 dim{1 size};
    Event 2 step, 4 graph_def - no wall_time, as the reference writes it) and the projector's three text files, byte for byte."""
    env = dict(os.environ, T4_SEED="1", T4_TB_FIXED_TIME=str(int(T0)))
    r = subprocess.run([binary, "-t", str(tmp_path), "-r", "g1"], input=GRAPH_SCRIPT, capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "check TensorBoard param" not in r.stdout and "not written" not in r.stdout, r.stdout
    run = os.path.join(str(tmp_path), "g1")
    recs = records(open(glob.glob(os.path.join(run, "events.out.tfevents.*"))[0], "rb").read())
    OP = {"conv2d ": "Conv2D", "maxpool": "MaxPool", "relu   ": "Relu", "flatten": "Reshape", "linear ": "MatMul", "softmax": "Softmax", "output ": "Output"}
    layers = [("conv2d ", (2, 8, 8, 1)), ("maxpool", (2, 8, 8, 3)), ("relu   ", (2, 4, 4, 3)), ("flatten", (2, 4, 4, 3)), ("linear ", (2, 1, 48, 1)),
              ("softmax", (2, 1, 5, 1)), ("output ", (2, 1, 5, 1))]

    def attrs(shape):
        dims = b"".join(f_len(2, f_i64(1, d)) for d in shape)
        return f_len(5, f_len(1, b"dtype") + f_len(2, f_i64(6, 1))) + f_len(5, f_len(1, b"shape") + f_len(2, f_len(7, dims)))
    names = ["%s_%d/%s" % (OP[l], i, l) for i, (l, _s) in enumerate(layers)]
    graph = f_len(1, f_len(1, b"input") + f_len(2, b"Placeholder") + attrs(layers[0][1]))
    for i, (l, shp) in enumerate(layers):
        graph += f_len(1, f_len(1, names[i].encode()) + f_len(2, OP[l].encode()) + f_len(3, (names[i - 1] if i else "input").encode()) + attrs(shp))
    want = [f_f64(1, T0) + f_i64(2, 0) + f_len(3, b"brain.Event:2"), f_i64(2, 0) + f_len(4, graph)]
    assert len(recs) == 2
    for i, (g, w) in enumerate(zip(recs, want)):
        assert g == w, "record %d differs:\n got %s\nwant %s" % (i, g.hex(), w.hex())
    # projector files: rows = samples, tab-separated, C++ `ostream << float` text; tag escaped like the run id
    assert open(os.path.join(run, "act_l_1_tensors.tsv")).read() == "0.5\t-1\t2\t0.25\n1\t1\t1\t1\n0\t0\t0\t1e-07\n"
    assert open(os.path.join(run, "act_l_1_metadata.tsv")).read() == "act_l_1.0\nact_l_1.1\nact_l_1.2\n"
    assert open(os.path.join(run, "w_tensors.tsv")).read() == "1\t2\t3\t4\t5\t6\n"        # a matrix is one sample (N = 1) of H*W*C values
    cfg = "".join('embeddings {\n  tensor_name: "%s"\n  tensor_path: "%s/%s_tensors.tsv"\n  metadata_path: "%s/%s_metadata.tsv"\n}\n' % (t, run, t, run, t) for t in ("act_l_1", "w"))
    assert open(os.path.join(run, "projector_config.pbtxt")).read() == cfg


def test_graph_event_and_projector_files_are_byte_exact(oracle_vm, tmp_path):
    _graph_and_embed(oracle_vm, tmp_path)


@pytest.mark.gpu
def test_graph_event_and_projector_files_from_the_product_vm(tmp_path):
    _graph_and_embed(TEN4, tmp_path)


def test_words_only_hint_without_a_log_directory(oracle_vm):
    env = {k: v for k, v in os.environ.items() if not k.startswith("T4_TB_")}
    r = subprocess.run([oracle_vm], input='0.5 s" x" .scalar\n3 .tbstep\nbye\n', capture_output=True, text=True, env=dict(env, T4_SEED="1"), timeout=60)
    # the reference's texts (System::_process_tb sys.cpp:234-253, confirmed on its own VM by tests/test_refhost_parity.py): tagged ops echo op / n / i / tag,
    # the untagged ones (.tbstep, .graph) add the hint
    assert "  sys#tbx(op=2, n=0.5, i=0, tag=x)\n" in r.stdout
    assert "  sys#tbx(op=1, n=0, i=3), check TensorBoard param -tlogdir -rrun_id\n" in r.stdout
