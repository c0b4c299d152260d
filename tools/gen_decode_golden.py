"""Build-container only: frames from the REAL libzstd 1.5.7 (tools/zref.py) that exercise the parts of
the format the pack path at level 12 never emits -- other levels, the content checksum, frames without a
content size (window descriptor), skippable frames, concatenated frames -- for the `unpack` decoder
(squishrs_amd/host/zstd_frame_decoder.hpp).  Writes tests/golden/zstd_decode_frames.json:
{name, kind, seed, nbytes, level, checksum, content_size, frame(hex), sha256(content)}.
The generating inputs come from tools/corpus.py, so the tests rebuild the expected content themselves."""
import ctypes
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus
import zref

ZSTD_c_compressionLevel, ZSTD_c_contentSizeFlag, ZSTD_c_checksumFlag = 100, 200, 201


def compress2(data: bytes, level: int, checksum: int, content_size: int) -> bytes:
    z = zref.lib()
    c = z.ZSTD_createCCtx()
    for k, v in ((ZSTD_c_compressionLevel, level), (ZSTD_c_checksumFlag, checksum), (ZSTD_c_contentSizeFlag, content_size)):
        assert not z.ZSTD_isError(z.ZSTD_CCtx_setParameter(c, k, v))
    cap = z.ZSTD_compressBound(len(data))
    dst = ctypes.create_string_buffer(cap)
    n = z.ZSTD_compress2(c, dst, cap, data, len(data))
    assert not z.ZSTD_isError(n)
    z.ZSTD_freeCCtx(c)
    return dst.raw[:n]


def main():
    cases = []
    plan = [("text", 11, 30000), ("records", 12, 24000), ("runs", 13, 40000), ("skewed", 14, 20000),
            ("mixed", 15, 150000), ("random", 16, 5000), ("text", 17, 700), ("text", 18, 140000)]
    for kind, seed, n in plan:
        data = corpus.make(kind, seed, n)
        for level in (1, 3, 7, 19, -5):
            for checksum, csize in ((0, 1), (1, 1), (0, 0)):
                if level in (7, -5) and (checksum, csize) != (0, 1):
                    continue
                if n > 100000 and (level, checksum, csize) not in ((1, 0, 1), (19, 1, 1), (19, 0, 0)):
                    continue
                if kind == "mixed" and (level, checksum, csize) != (19, 1, 1):
                    continue
                fr = compress2(data, level, checksum, csize)
                cases.append({"name": f"{kind}-{seed}-{n}-L{level}-c{checksum}-s{csize}", "kind": kind, "seed": seed,
                              "nbytes": n, "level": level, "checksum": checksum, "content_size": csize,
                              "frame": fr.hex(), "sha256": hashlib.sha256(data).hexdigest()})
    # two frames back to back with a skippable frame between them
    a, b = corpus.make("text", 21, 5000), corpus.make("records", 22, 6000)
    skip = bytes.fromhex("502a4d18") + (7).to_bytes(4, "little") + b"skipped"
    fr = compress2(a, 3, 1, 1) + skip + compress2(b, 19, 0, 0)
    cases.append({"name": "concat-skippable", "kind": "concat", "parts": [["text", 21, 5000], ["records", 22, 6000]],
                  "frame": fr.hex(), "sha256": hashlib.sha256(a + b).hexdigest()})
    # empty content
    cases.append({"name": "empty", "kind": "concat", "parts": [], "frame": compress2(b"", 12, 0, 1).hex(),
                  "sha256": hashlib.sha256(b"").hexdigest()})
    out = os.path.join(ROOT, "tests", "golden", "zstd_decode_frames.json")
    json.dump({"source": "libzstd 1.5.7 via tools/gen_decode_golden.py", "cases": cases}, open(out, "w"), indent=0)
    print(len(cases), "cases,", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
