"""Minimal PNG / PNM reader-writer for the tests (no PIL dependency): 8-bit gray or RGB, non-interlaced."""
import struct
import zlib

import numpy as np


def write_png(path, arr, filter_type=0):
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else 3
    rows = a.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        cur = rows[y]
        if filter_type == 0:
            f = cur
        elif filter_type == 1:
            f = (cur - np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])) & 255
        elif filter_type == 2:
            f = (cur - prev) & 255
        else:
            raise ValueError
        raw.append(filter_type)
        raw += f.astype(np.uint8).tobytes()
        prev = cur

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as fo:
        fo.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0 if ch == 1 else 2, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


def read_png(path):
    buf = open(path, "rb").read()
    assert buf[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat = 8, b""
    while pos < len(buf):
        n, tag = struct.unpack(">I4s", buf[pos:pos + 8])
        d = buf[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", d[:10])
        elif tag == b"IDAT":
            idat += d
        pos += 12 + n
    ch = {0: 1, 2: 3}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    assert depth == 8 and np.all(raw[:, 0] == 0), "test reader handles filter 0 only"
    out = raw[:, 1:].reshape(h, w, ch)
    return out[..., 0] if ch == 1 else out


def write_pnm(path, arr):
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n" if a.ndim == 2 else b"P6\n")
        f.write(f"{a.shape[1]} {a.shape[0]}\n255\n".encode())
        f.write(a.tobytes())


def read_pnm(path):
    buf = open(path, "rb").read()
    parts = buf.split(None, 4)
    magic, w, h, mx = parts[0], int(parts[1]), int(parts[2]), int(parts[3])
    off = len(buf) - w * h * (3 if magic == b"P6" else 1)
    a = np.frombuffer(buf[off:], np.uint8)
    return a.reshape(h, w, 3) if magic == b"P6" else a.reshape(h, w)
