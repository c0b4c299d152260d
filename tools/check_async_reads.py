"""Guard for the hand-issued LDS reads of stage 1's filter (sq_zstd_fast.h, `round`): the 18 `ds_read_u8` of a round
are inline asm, their single `s_waitcnt lgkmcnt(0)` is inline asm too, and hipcc does not know that the destination
registers are still in flight in between.  This script compiles the library to ISA and fails if ANY instruction between
a group's first read and its wait names one of the group's destination registers (a copy or a spill there would move
garbage).  Run by tests/test_abi_cpu.py; usage: python tools/check_async_reads.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


MAX_SPAN = 120  # lines between a group's first read and its wait (18 reads + 18 bit indices + a few address moves today)


def _vregs(text):
    """Every vector register an instruction names: single operands (v17) and the members of ranges (v[8:11], as in
    dwordx2 / dwordx4 memory operations, 64-bit moves, packed math, MFMA / accvgpr moves)."""
    t = text.split(";")[0]
    out = set(re.findall(r"\bv\d+\b", t))
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t):
        out.update(f"v{k}" for k in range(int(a), int(b) + 1))
    return out


def scan(asm_text):
    lines = asm_text.split("\n")
    groups, hazards, i = 0, [], 0
    while i < len(lines):
        if lines[i].strip().startswith("ds_read_u8") and lines[i - 1].strip() == ";;#ASMSTART":
            j, regs = i, set()
            while not (lines[j].strip() == "s_waitcnt lgkmcnt(0)" and lines[j - 1].strip() == ";;#ASMSTART"):
                t = lines[j].strip()
                if t.startswith("ds_read_u8") and lines[j - 1].strip() == ";;#ASMSTART":
                    regs.add(t.split()[1].rstrip(","))
                elif t and not t.startswith(";") and not t.startswith("."):
                    if t.endswith(":") or t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("s_endpgm"):
                        hazards.append((j + 1, "control flow between the reads and their wait: " + t))
                    if _vregs(t) & regs:
                        hazards.append((j + 1, t))
                j += 1
                if j >= len(lines) or j - i > MAX_SPAN:
                    hazards.append((i + 1, f"reads without their s_waitcnt within {MAX_SPAN} lines"))
                    break
            if len(regs) != 18:
                hazards.append((i + 1, f"a group of {len(regs)} reads (18 expected)"))
            groups += 1
            i = j
        i += 1
    return groups, hazards


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "sq_api.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                               os.path.join(ROOT, "squishrs_amd", "csrc", "sq_api.hip")] + sys.argv[1:],
                              stderr=subprocess.DEVNULL)
        groups, hazards = scan(open(out).read())
    print(f"{groups} groups of hand-issued reads, {len(hazards)} hazards")
    for ln, t in hazards[:20]:
        print(f"  line {ln}: {t}")
    return 0 if groups >= 10 and not hazards else 1


if __name__ == "__main__":
    sys.exit(main())
