"""Writes a list of RGB frames as a Motion-JPEG AVI -- what the reference's AgarioEnv.generate_video produces with
cv2.VideoWriter(fourcc 'MJPG', 60 fps) (/root/reference/gym_agario/AgarioEnv.py:379-400).  cv2 is used when it is importable; otherwise
the same container is written here (RIFF / AVI 1.0 with an idx1 index, one JPEG per '00dc' chunk, JPEG encoding by Pillow)."""
import io
import struct

import numpy as np


def _chunk(fourcc, payload):
    return fourcc + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")


def _list(kind, payload):
    return b"LIST" + struct.pack("<I", len(payload) + 4) + kind + payload


def write_mjpeg_avi(path, frames, fps=60.0, quality=90):
    frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
    if not frames:
        raise ValueError("no frames")
    h, w = frames[0].shape[:2]
    try:
        import cv2
        video = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"MJPG"), float(fps), (w, h))
        if not video.isOpened():
            raise RuntimeError("Error: VideoWriter failed to open.")
        for f in frames:
            video.write(cv2.cvtColor(f, cv2.COLOR_RGB2BGR))
        video.release()
        return "cv2"
    except ImportError:
        pass
    from PIL import Image
    jpegs = []
    for f in frames:
        if f.shape[:2] != (h, w):
            raise ValueError("frames differ in size")
        b = io.BytesIO(); Image.fromarray(f[..., :3], "RGB").save(b, format="JPEG", quality=quality); jpegs.append(b.getvalue())
    n, us = len(jpegs), int(round(1e6 / fps))
    biggest = max(len(j) for j in jpegs)
    avih = struct.pack("<IIIIIIIIII4I", us, int(biggest * fps), 0, 0x10, n, 0, 1, biggest, w, h, 0, 0, 0, 0)          # AVIF_HASINDEX
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, 1, int(round(fps)), 0, n, biggest, 0xFFFFFFFF, 0, 0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)                                # BITMAPINFOHEADER
    hdrl = _list(b"hdrl", _chunk(b"avih", avih) + _list(b"strl", _chunk(b"strh", strh) + _chunk(b"strf", strf)))
    movi_payload, index, off = b"", b"", 4
    for j in jpegs:
        c = _chunk(b"00dc", j)
        index += b"00dc" + struct.pack("<III", 0x10, off, len(j))   # AVIIF_KEYFRAME, offset from the 'movi' tag
        movi_payload += c; off += len(c)
    body = hdrl + _list(b"movi", movi_payload) + _chunk(b"idx1", index)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body) + 4) + b"AVI " + body)
    return "builtin"
