"""GPU tier: the reference's entry points (main.rs:8-45) end to end: scene build, render, resolve, PNG on disk."""
import struct
import zlib

import numpy as np
import pytest

from raytracing_rust_amd import __main__ as cli


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, size = 8, b"", None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(kind + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if kind == b"IHDR":
            w, h, depth, color = struct.unpack(">IIBB", body[:10])
            assert (depth, color) == (8, 2)
            size = (w, h)
        elif kind == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(size[1], 1 + size[0] * 3)
    assert (raw[:, 0] == 0).all()                      # filter type 0 on every row
    return raw[:, 1:].reshape(size[1], size[0], 3)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,width,spp", [("quick", 96, 16), ("cornell", 80, 25), ("smoke", 64, 16)])
def test_entry_points_write_the_image(tmp_path, scene, width, spp, capsys):
    out = tmp_path / f"{scene}.png"
    assert cli.main([scene, "--width", str(width), "--spp", str(spp), "--depth", "10", "--out", str(out)]) == 0
    img = read_png(out)
    assert img.shape == (width, width, 3)
    assert img.std() > 5 and 5 < img.mean() < 250      # a picture, not a constant
    assert "Msamples/s" in capsys.readouterr().out


def test_usage_without_arguments(capsys):
    assert cli.main([]) == 2
    assert "cornell|final|quick" in capsys.readouterr().err
