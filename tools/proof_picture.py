"""Renders the default frame with a diagnostic build of the library (-DRM_DIAG_PROOF: every pixel shows
what the shadow-ray proof said at the primary hit) and writes it as a PNG.

usage: RM_LIB=path/to/librm_diagproof.so python tools/proof_picture.py out.png [width height]
"""
import os
import struct
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import _lib, default_scene

if os.environ.get("RM_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["RM_LIB"])
from raymarching_amd.engine import Engine

out = sys.argv[1]
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (960, 540)
e = Engine(default_scene(), W, H)
e.compute_lightmaps(4)
e.input(0, 0, (0, 0, 0), 0)
buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
e.render_rows_device(buf.data_ptr(), 0, H)
torch.cuda.synchronize()
rgb = buf[:, :, [2, 1, 0]].cpu().numpy()
counts = {}
for name, colour_test in (("red", rgb[:, :, 0] > 200), ("green", (rgb[:, :, 1] > 200) & (rgb[:, :, 2] < 100)), ("blue", rgb[:, :, 2] > 200)):
    counts[name] = float(colour_test.mean())
print("share of pixels: marched %.3f, proven in a mixed wave %.3f, whole wave proven %.3f" % (counts["red"], counts["green"], counts["blue"]))
raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(H))


def chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


with open(out, "wb") as f:
    f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
