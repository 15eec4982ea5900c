"""Minimal float32 TIFF / EDF image I/O (stand-in for CodePython/InputOutput/pagailleIO.py, which wraps fabio).

fabio is not available here, so the two formats the driver writes (main.py:99-110, `saving_format` '.tif' or '.edf')
are produced directly: baseline uncompressed little-endian TIFF with one float32 sample per pixel, and ESRF EDF
(ASCII header padded to 512-byte blocks + raw little-endian float32).  Both round-trip through openImage.
"""
import os
import struct

import numpy as np


def _as_f32(data):
    if hasattr(data, "detach"):
        data = data.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(data), dtype="<f4")


def save_tif_image(data, filename):
    img = _as_f32(data)
    if img.ndim != 2:
        raise ValueError("save_tif_image expects a 2-D image")
    h, w = img.shape
    header = struct.pack("<2sHI", b"II", 42, 8)
    tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, 32), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 0),
            (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, img.nbytes), (339, 3, 1, 3)]
    ifd_size = 2 + 12 * len(tags) + 4
    data_off = 8 + ifd_size
    ifd = struct.pack("<H", len(tags))
    for tag, typ, cnt, val in tags:
        if tag == 273:
            val = data_off
        ifd += struct.pack("<HHII", tag, typ, cnt, val) if typ == 4 else struct.pack("<HHIHH", tag, typ, cnt, val, 0)
    ifd += struct.pack("<I", 0)
    with open(filename, "wb") as f:
        f.write(header + ifd + img.tobytes())


def saveEdf(data, filename):
    img = _as_f32(data)
    h, w = img.shape
    # ESRF data format, the header grammar fabio's EdfImage writes: "{\n", lines "key = value ;\n", space padding, "}\n", the
    # whole header a multiple of 512 bytes; Dim_1 is the FAST axis (columns); raw little-endian float32 follows
    hdr = ("{\nHeaderID = EH:000001:000000:000000 ;\nImage = 1 ;\nByteOrder = LowByteFirst ;\nDataType = FloatValue ;\n"
           "Dim_1 = %d ;\nDim_2 = %d ;\nSize = %d ;\n" % (w, h, img.nbytes))
    pad = (-(len(hdr) + 2)) % 512
    hdr = hdr + " " * pad + "}\n"
    with open(filename, "wb") as f:
        f.write(hdr.encode("ascii") + img.tobytes())


def save_image(data, filename):
    """Dispatch on the extension like pagailleIO.py:125-129."""
    ext = os.path.splitext(filename)[1].lower()
    if ext in (".tif", ".tiff"):
        save_tif_image(data, filename)
    elif ext == ".edf":
        saveEdf(data, filename)
    elif ext == ".npy":
        np.save(filename, _as_f32(data))
    else:
        raise ValueError("unknown image format %r" % ext)


_EDF_TYPES = {"FloatValue": "f4", "Float": "f4", "DoubleValue": "f8", "UnsignedShort": "u2", "SignedShort": "i2",
              "UnsignedInteger": "u4", "SignedInteger": "i4", "UnsignedByte": "u1", "SignedByte": "i1",
              "UnsignedLong": "u4", "SignedLong": "i4"}


def _open_edf(raw):
    """Any single-frame EDF: header = "{" ... "}\n" (padded to 512-byte blocks), key = value ; lines."""
    end = raw.index(b"}\n") + 2
    fields = {}
    for line in raw[:end].decode("ascii", "replace").split(";"):
        if "=" in line:
            k, v = line.split("=", 1)
            fields[k.strip().lstrip("{").strip()] = v.strip()
    w, h = int(fields["Dim_1"]), int(fields["Dim_2"])
    order = "<" if fields.get("ByteOrder", "LowByteFirst") == "LowByteFirst" else ">"
    dt = np.dtype(order + _EDF_TYPES[fields.get("DataType", "FloatValue")])
    return np.frombuffer(raw, dtype=dt, count=w * h, offset=end).reshape(h, w).astype(np.float32 if dt.kind == "f" else dt.newbyteorder("="))


def openImage(filename):
    """pagailleIO.py:25-47: EDF / TIFF -> 2-D array.  TIFF files written by other tools (strips, big-endian, integer
    samples) go through PIL when it is importable; the files this module writes are also read without it."""
    ext = os.path.splitext(filename)[1].lower()
    if ext == ".npy":
        return np.load(filename)
    raw = open(filename, "rb").read()
    if ext == ".edf":
        return _open_edf(raw)
    try:
        from PIL import Image
        with Image.open(filename) as im:
            return np.array(im)
    except ImportError:
        pass
    if raw[:4] != b"II*\x00":
        raise ValueError("without PIL only little-endian baseline TIFF written by save_tif_image is supported")
    (ifd_off,) = struct.unpack_from("<I", raw, 4)
    (n,) = struct.unpack_from("<H", raw, ifd_off)
    vals = {}
    for i in range(n):
        tag, typ, cnt = struct.unpack_from("<HHI", raw, ifd_off + 2 + 12 * i)
        vals[tag] = struct.unpack_from("<I" if typ == 4 else "<H", raw, ifd_off + 2 + 12 * i + 8)[0]
    return np.frombuffer(raw, dtype="<f4", count=vals[256] * vals[257], offset=vals[273]).reshape(vals[257], vals[256]).copy()
