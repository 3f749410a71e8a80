"""The two PNG writers (the display stage of SURVEY.md §8(f) rank 1): the dependency-free one of
the C++ host mirror (host/raytracer.hpp: stored deflate blocks) and the Python helper.  A small
decoder below checks signature, chunk CRCs, the zlib stream and the pixels."""
import os
import struct
import subprocess
import zlib

import numpy as np

from conftest import ROOT


def decode_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xffffffff, tag
        chunks.append((tag, body))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, flt, il = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, flt, il) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, w * 3 + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def test_python_png_roundtrip(rt, tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    rt.save_png(str(tmp_path / "a.png"), img)
    assert np.array_equal(decode_png(tmp_path / "a.png"), img)
    frame = np.array([[[0.0, 0.5, 1.0], [2.0, -1.0, 0.999]]], np.float32)       # src/main.cu:343-371: int(px*255), clamped
    rt.save_png(str(tmp_path / "b.png"), frame)
    assert decode_png(tmp_path / "b.png").tolist() == [[[0, 127, 255], [255, 0, 254]]]


def test_cpp_png_writer(tmp_path):
    """more than one 65,535-byte stored block, odd sizes"""
    src = tmp_path / "t.cpp"
    src.write_text('#include "raytracer.hpp"\n'
                   'int main(int argc, char **argv) { int w = 301, h = 173; std::vector<float> f((size_t)w * h * 3);\n'
                   '  for (int i = 0; i < w * h; i++) { f[3 * i] = (i % w) / 300.0f; f[3 * i + 1] = (i / w) / 172.0f; f[3 * i + 2] = ((i * 7) % 256) / 255.0f; }\n'
                   '  rtamd::write_png(argv[1], rtamd::parse_pixel_colours(f, w, h), w, h); return 0; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "ray-tracer_amd", "host"), str(src), "-o", str(exe)])
    subprocess.check_call([str(exe), str(tmp_path / "c.png")])
    got = decode_png(tmp_path / "c.png")
    i = np.arange(301 * 173)
    want = np.stack([((i % 301).astype(np.float32) / np.float32(300.0) * np.float32(255)).astype(np.int64),
                     ((i // 301).astype(np.float32) / np.float32(172.0) * np.float32(255)).astype(np.int64),
                     (((i * 7) % 256).astype(np.float32) / np.float32(255.0) * np.float32(255)).astype(np.int64)], axis=1).clip(0, 255).astype(np.uint8).reshape(173, 301, 3)
    assert np.array_equal(got, want)
