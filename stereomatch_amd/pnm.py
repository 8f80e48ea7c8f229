"""PGM / PPM / PMF reader and writer with the reference's in-memory conventions.

ReadFilePGM (ImageIO.cpp:333-402): P5 -> 1 band; P6 -> 4 bands stored B,G,R,A with
A = 255 (:390-396).  WriteFilePGM (ImageIO.cpp:406-486): header "P5\\n%d %d\\n255\\n"
(:431-432); 4-band images are written as P6 in R,G,B order without alpha (:456-478).
"""
import numpy as np


def _read_header(f, magic):
    # read_header, ImageIO.cpp:301-330: magic, optional '#' comment lines, three ints,
    # then exactly one newline (optionally preceded by CR).
    if f.read(2) != magic:
        raise ValueError("wrong magic code")
    vals = []
    tok = b""
    while len(vals) < 3:
        c = f.read(1)
        if not c:
            raise ValueError("truncated header")
        if c == b"#" and not tok:
            while f.read(1) not in (b"\n", b""):
                pass
        elif c in b" \t\r\n":
            if tok:
                vals.append(int(tok))
                tok = b""
                if len(vals) == 3 and c == b"\r":
                    f.read(1)
        else:
            tok += c
    return vals


def read_image(path):
    """Returns uint8 [H, W, nBands] with nBands = 1 (.pgm) or 4 (.ppm, BGRA), or float32
    [H, W, nBands] for a .pmf (the reference's multi-band float container, e.g. a DSI)."""
    with open(path, "rb") as f:
        if path.endswith(".pgm"):
            w, h, _ = _read_header(f, b"P5")
            data = np.frombuffer(f.read(w * h), np.uint8)
            if data.size != w * h:
                raise ValueError("%s: file is too short" % path)
            return data.reshape(h, w, 1).copy()
        if path.endswith(".ppm"):
            w, h, _ = _read_header(f, b"P6")
            data = np.frombuffer(f.read(w * h * 3), np.uint8)
            if data.size != w * h * 3:
                raise ValueError("%s: file is too short" % path)
            rgb = data.reshape(h, w, 3)
            out = np.empty((h, w, 4), np.uint8)
            out[..., 0] = rgb[..., 2]
            out[..., 1] = rgb[..., 1]
            out[..., 2] = rgb[..., 0]
            out[..., 3] = 255
            return out
        if path.endswith(".pmf"):
            # P9: width, height, bands, then rows of native floats (ImageIO.cpp:352-376)
            w, h, b = _read_header(f, b"P9")
            data = np.frombuffer(f.read(w * h * b * 4), np.float32)
            if data.size != w * h * b:
                raise ValueError("%s: file is too short" % path)
            return data.reshape(h, w, b).copy()
    raise ValueError("%s: unsupported extension" % path)


def pgm_bytes(img):
    """Bytes of the file WriteFilePGM would write for a 1-band uint8 image."""
    img = np.asarray(img, np.uint8)
    if img.ndim == 3:
        img = img[..., 0]
    h, w = img.shape
    return b"P5\n%d %d\n255\n" % (w, h) + np.ascontiguousarray(img).tobytes()


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(pgm_bytes(img))


def pmf_bytes(img):
    """Bytes of the file WriteFilePGM writes for a float image (ImageIO.cpp:428-441)."""
    img = np.ascontiguousarray(img, np.float32)
    if img.ndim == 2:
        img = img[..., None]
    h, w, b = img.shape
    return b"P9\n%d %d\n%d\n" % (w, h, b) + img.tobytes()


def write_pmf(path, img):
    with open(path, "wb") as f:
        f.write(pmf_bytes(img))
