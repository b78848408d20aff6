"""A PNG writer for 8-bit RGB images (the renderer writes its results through png++, libgi/framebuffer.cpp:21-30,
driver/main.cpp:29-33; this is the same file format from the standard library's zlib)."""
import struct
import zlib


def write_rgb8(path, rgb):
    """rgb: uint8 array (h, w, 3), top row first."""
    h, w, c = rgb.shape
    if c != 3 or rgb.dtype.itemsize != 1:
        raise ValueError("expected an (h, w, 3) uint8 array")

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))  # filter type 0 on every scanline
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def read_rgb8(path):
    """Reads back what write_rgb8 wrote (8-bit RGB, filter 0 only); for tests."""
    import numpy as np
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, w, h, idat = 8, 0, 0, b""
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    if raw[:, 0].any():
        raise ValueError("only filter type 0 is supported")
    return raw[:, 1:].reshape(h, w, 3).copy()
