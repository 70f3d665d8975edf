"""A container format for batches and tiled images.

The reference writes five fixed-name files per compress() call and overwrites them for every image and
every tile (model.py:226-249, inference_high_resolution.py:246), so nothing but the last tile of the last
image survives on disk.  This container keeps every entry decodable:

    magic  "CGIC"  | u16 version = 1 | u16 flags = 0 | u32 n_entries
    n_entries x entry header (44 bytes, little endian):
        u32 image_id | u32 y | u32 x | u32 height | u32 width   (pixel rectangle in the padded image)
        u8 mode | 3 pad bytes | 5 x i32 stream length (-1 = stream not written in this mode)
    payload: the streams of entry 0 (in stream order), entry 1, ...  byte-identical to the reference's .bin files

`write_legacy` in codec.py still produces the reference's own five files for one image.
"""
import struct

from ._lib import STREAM_NAMES

MAGIC = b"CGIC"
VERSION = 1
_HDR = struct.Struct("<4sHHI")
_ENT = struct.Struct("<IIIIIB3x5i")


def pack(entries):
    """entries: list of dict(image_id, y, x, height, width, mode, streams={name: bytes}) -> bytes"""
    head = [_HDR.pack(MAGIC, VERSION, 0, len(entries))]
    body = []
    for e in entries:
        lens = [len(e["streams"][n]) if n in e["streams"] else -1 for n in STREAM_NAMES]
        head.append(_ENT.pack(e["image_id"], e["y"], e["x"], e["height"], e["width"], e["mode"], *lens))
        body.extend(e["streams"][n] for n in STREAM_NAMES if n in e["streams"])
    return b"".join(head + body)


def unpack(blob):
    magic, version, _flags, n = _HDR.unpack_from(blob, 0)
    if magic != MAGIC or version != VERSION:
        raise ValueError("not a CGIC container (or unknown version)")
    off = _HDR.size
    metas = []
    for _ in range(n):
        metas.append(_ENT.unpack_from(blob, off))
        off += _ENT.size
    out = []
    for image_id, y, x, hh, ww, mode, *lens in metas:
        streams = {}
        for name, ln in zip(STREAM_NAMES, lens):
            if ln >= 0:
                if off + ln > len(blob):
                    raise ValueError("truncated container")
                streams[name] = bytes(blob[off:off + ln])
                off += ln
        out.append(dict(image_id=image_id, y=y, x=x, height=hh, width=ww, mode=mode, streams=streams))
    if off != len(blob):
        raise ValueError("trailing bytes after the last stream")
    return out


def entries_from_batch(comp, height, width, first_image_id=0):
    """CompressedBatch of whole images -> container entries"""
    return [dict(image_id=first_image_id + b, y=0, x=0, height=height, width=width, mode=comp.mode, streams=s)
            for b, s in enumerate(comp.to_host())]


def entries_from_tiled(tiled, image_id=0):
    """TiledImage -> container entries, row-major tile order"""
    modes = [None] * len(tiled.tiles)
    for idxs, comp, _ in tiled.groups:
        for i in idxs:
            modes[i] = comp.mode
    return [dict(image_id=image_id, y=y, x=x, height=th, width=tw, mode=modes[i], streams=s)
            for i, ((y, x, th, tw), s) in enumerate(zip(tiled.tiles, tiled.streams()))]


def bits_per_pixel(entries, image_hw):
    """sum of stream bytes * 8 / (H * W) -- equals the reference's bpp accounting for whole images and tiles"""
    return sum(len(v) for e in entries for v in e["streams"].values()) * 8 / (image_hw[0] * image_hw[1])
