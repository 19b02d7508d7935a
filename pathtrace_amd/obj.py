"""LoadObj / DecomposeObjects (obj_loader.go:19-111, renderer.go:293-324) through the C ABI's asset loader
(csrc/obj_load.cpp), plus the image decoding LoadObj does with Go's image package (obj_loader.go:29-57):
binary PPM/PGM and 8-bit non-interlaced PNG, packed as CacheImage packs texels (renderer.go:94-119)."""
from __future__ import annotations

import ctypes as C
import os
import struct
import zlib

import numpy as np

from . import _lib
from ._lib import PtmiError
from .scenes import SceneData, pack_rgba


def read_image(path) -> np.ndarray:
    """-> [h,w] u32 packed R | G<<8 | B<<16 | A<<24."""
    data = open(path, "rb").read()
    if data[:2] in (b"P6", b"P5"):
        toks, pos = [], 2
        while len(toks) < 3:                      # width height maxval, '#' comments allowed
            while data[pos:pos + 1].isspace():
                pos += 1
            if data[pos:pos + 1] == b"#":
                pos = data.index(b"\n", pos) + 1
                continue
            end = pos
            while not data[end:end + 1].isspace():
                end += 1
            toks.append(int(data[pos:end])); pos = end
        pos += 1
        w, h, mx = toks
        if mx != 255:
            raise ValueError(f"{path}: only maxval 255 is supported")
        ch = 3 if data[:2] == b"P6" else 1
        px = np.frombuffer(data, np.uint8, w * h * ch, pos).reshape(h, w, ch)
        return pack_rgba(np.repeat(px, 3, -1) if ch == 1 else px)
    if data[:8] == b"\x89PNG\r\n\x1a\n":
        pos, idat, hdr = 8, b"", None
        while pos < len(data):
            n, typ = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            if typ == b"IHDR":
                hdr = struct.unpack(">IIBBBBB", body)
            elif typ == b"IDAT":
                idat += body
            pos += 12 + n
        w, h, depth, ctype, _, _, interlace = hdr
        nch = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
        if depth != 8 or nch is None or interlace:
            raise ValueError(f"{path}: only 8-bit non-interlaced grey/RGB(A) PNG is supported")
        raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * nch)
        out = np.zeros((h, w * nch), np.uint8)
        prev = np.zeros(w * nch, np.int32)
        for y in range(h):                        # PNG filters, bytes per pixel = nch
            ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
            if ft == 0:
                cur = line
            elif ft == 2:
                cur = (line + prev) & 255
            else:
                cur = np.zeros_like(line)
                for x in range(w * nch):
                    a = cur[x - nch] if x >= nch else 0
                    b = prev[x]
                    c = prev[x - nch] if x >= nch else 0
                    if ft == 1:
                        pred = a
                    elif ft == 3:
                        pred = (a + b) >> 1
                    else:
                        p0 = a + b - c
                        pa, pb, pc = abs(p0 - a), abs(p0 - b), abs(p0 - c)
                        pred = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                    cur[x] = (line[x] + pred) & 255
            out[y] = cur
            prev = cur
        px = out.reshape(h, w, nch)
        if nch == 1:
            px = np.repeat(px, 3, -1)
        elif nch == 2:
            px = np.concatenate([np.repeat(px[..., :1], 3, -1), px[..., 1:]], -1)
        return pack_rgba(px)
    raise ValueError(f"{path}: unsupported image format (PPM, PGM and PNG are)")


def LoadObj(path, scale_factor=1.0, position=(0.0, 0.0, 0.0), load_textures=True) -> SceneData:
    """obj_loader.go:19 + renderer.go:293: the arrays of one object, ready for Renderer / ptmi_upload_*.
    Lights, sky and camera are not part of an OBJ; the caller sets them on the returned SceneData."""
    lib = _lib.load()
    h = C.c_void_p()
    pos = (C.c_float * 3)(*[float(x) for x in position])
    rc = lib.ptmi_obj_load(os.fsencode(path), C.c_float(scale_factor), pos, C.byref(h))
    if rc:
        raise PtmiError(rc, lib.ptmi_obj_last_error().decode())
    try:
        nv, nt, nm, nw = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        has_uv = C.c_int32()
        lib.ptmi_obj_sizes(h, C.byref(nv), C.byref(nt), C.byref(nm), C.byref(has_uv), C.byref(nw))
        verts = np.zeros((nv.value, 3), np.float32); tris = np.zeros((nt.value, 3), np.uint32)
        nrm = np.zeros((3 * nt.value, 3), np.float32); tmat = np.zeros(nt.value, np.uint32)
        mats = np.zeros((nm.value, 16), np.float32)
        uvs = np.zeros((nt.value, 6), np.float32) if has_uv.value else None
        F, U = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        lib.ptmi_obj_get(h, verts.ctypes.data_as(F), tris.ctypes.data_as(U), nrm.ctypes.data_as(F), tmat.ctypes.data_as(U),
                         mats.ctypes.data_as(C.c_void_p), uvs.ctypes.data_as(F) if uvs is not None else None)
        names = [lib.ptmi_obj_string(h, 0, i).decode() for i in range(nm.value)]
        map_kd = [lib.ptmi_obj_string(h, 1, i).decode() for i in range(nm.value)]
        map_bump = [lib.ptmi_obj_string(h, 2, i).decode() for i in range(nm.value)]
        warnings = [lib.ptmi_obj_string(h, 3, i).decode() for i in range(nw.value)]
        objdir = lib.ptmi_obj_string(h, 4, 0).decode()
    finally:
        lib.ptmi_obj_free(h)
    textures, cache = [], {}
    if load_textures and uvs is not None:
        mu = mats.view(np.uint32)
        for i in range(nm.value):
            for col, name in ((13, map_bump[i]), (12, map_kd[i])):      # bump first, like obj_loader.go:29
                if not name:
                    continue
                if name not in cache:
                    # the reference opens the name as written, relative to the working directory (obj_loader.go:35);
                    # fall back to the directory of the .obj, where such files normally live
                    cand = name if os.path.exists(name) else os.path.join(objdir, name)
                    textures.append(read_image(cand))
                    cache[name] = len(textures)
                mu[i, col] = cache[name]
    sc = SceneData(verts=verts, tris=tris, normals=nrm, tri_material=tmat, materials=mats,
                   lights=np.zeros((0, 12), np.float32), sky=np.zeros(16, np.float32), names=names,
                   uvs=uvs, textures=textures or None)
    sc.warnings = warnings
    sc.map_kd, sc.map_bump = map_kd, map_bump
    return sc
