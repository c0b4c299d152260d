"""Static instruction census of k_incompressible_batch's scan loops (CPU only: hipcc -S --cuda-device-only).
For every loop of the kernel that holds the filter's 18 ds_read_u8 groups it prints, per ROUND (a group = one round of
1 KiB), the instructions of the round's main block and of the whole loop body divided by the rounds in it, by class
(VALU / SALU / LDS / VMEM / v_readlane = SGPR-spill reloads).  The dynamic counts come from the SQ counters on the GPU
(tools/profile_sq.sh); this is the quick before/after check while editing the loop.
usage: python tools/isa_stage1.py [--keep file.s] [extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_isa(extra):
    out = os.path.join(tempfile.gettempdir(), "sq_api_stage1.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
           os.path.join(ROOT, "squishrs_amd", "csrc", "sq_api.hip"), "-Rpass-analysis=kernel-resource-usage"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(1)
    res = {}
    cur = None
    for line in r.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            res.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    return out, res


def klass(s):
    op = s.split()[0]
    if op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def main():
    extra = [a for a in sys.argv[1:] if a != "--keep"]
    keep = None
    if "--keep" in sys.argv:
        i = sys.argv.index("--keep")
        keep = sys.argv[i + 1]
        extra = [a for a in extra if a != keep]
    path, res = compile_isa(extra)
    text = open(path).read().split("\n")
    if keep:
        open(keep, "w").write("\n".join(text))
    # the kernel's body
    start = next(i for i, l in enumerate(text) if re.match(r"^_ZN3sqz22k_incompressible_batch\S*:", l))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start:end]
    for k, v in res.items():
        if "k_incompressible_batch" in k or "k_compress_batch" in k:
            print(k[:40], v)
    # blocks
    blocks = []
    cur = ["entry", None, []]
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            blocks.append(cur)
            hdr = None
            c = m.group(2) or ""
            mh = re.search(r"Header=(BB\d+_\d+)", c) or re.search(r"Parent Loop (BB\d+_\d+)", c)
            cur = [m.group(1), c, []]
            continue
        s = l.strip()
        if s and not s.startswith(";") and not s.startswith("."):
            cur[2].append(s)
        elif s.startswith(";") and cur[1] is not None and ("Loop" in s):
            cur[1] = (cur[1] or "") + " " + s
    blocks.append(cur)
    # loops at depth 2 (the scan loops): group blocks by the "Header=BBx Depth=2" they name, including inner loops'
    # blocks that follow (they name their own header; attribute by position between first and last block of the loop)
    names = [b[0] for b in blocks]
    loops = {}
    for i, b in enumerate(blocks):
        m = re.search(r"Header=(BB\d+_\d+) Depth=2", b[1] or "")
        if m:
            loops.setdefault(m.group(1), []).append(i)
    for h, idxs in loops.items():
        lo = names.index(".L" + h) if (".L" + h) in names else min(idxs)
        hi = max(idxs)
        span = blocks[lo:hi + 1]
        rounds = [b for b in span if sum(1 for s in b[2] if s.startswith("ds_read_u8")) >= 18]
        if not rounds:
            continue
        tot = {}
        for b in span:
            for s in b[2]:
                tot[klass(s)] = tot.get(klass(s), 0) + 1
        r = len(rounds)
        print(f"loop {h}: {r} rounds, {hi - lo + 1} blocks, {sum(len(b[2]) for b in span)} instructions "
              f"(~{sum(len(b[2]) for b in span) * 6 // 1024} KiB of code)")
        print("   whole loop body per round: " + "  ".join(f"{k} {v / r:.0f}" for k, v in sorted(tot.items())))
        for b in rounds:
            c = {}
            for s in b[2]:
                c[klass(s)] = c.get(klass(s), 0) + 1
            print(f"   main block {b[0]}: " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()
