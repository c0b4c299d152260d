"""Guard for the persistent work-queue loops of the two product kernels ("a hang from an inlined
per-chunk body"; CPU only: hipcc -S --cuda-device-only).

What the round-3 incident's ISA shows (profiles/round4_hang_isa/): hipcc lowers every wave-uniform branch as
`s_and_b64 vcc, exec, <cond>; s_cbranch_vccnz`, i.e. a wave-uniform `while (remaining)` loop can only be LEFT while EXEC
is not zero, and with a per-chunk body that contains out-of-line calls inlined into the queue loop, the queue's own state
(its lane-0 mask, the chunk counter, the list pointer) shares scalar registers and SGPR-spill lanes with that body and has
to be re-materialised after every call.  Out of line, the body's epilogue restores EXEC and the caller reloads its whole
SGPR state from a vector register it saved with all lanes enabled.  This script asserts the shape that has never hung:

  k_compress_batch       (1) the per-chunk body is reached through ONE s_swappc_b64 to compress_one and the kernel itself
                             holds no other call; (2) the queue's global_atomic_add sits inside the lane-election
                             s_and_saveexec region(s) only and EXEC is restored (s_or_b64 exec) before the next branch that
                             tests it; (3) the kernel body stays small (the block loop is not inlined into it).
  k_incompressible_batch its per-job body IS inlined (one workgroup = four waves that meet at barriers: it cannot be a
                             per-wave call), which is a different pattern from the incident's: (4) the only calls are to the
                             leaves fast_sched_mask_rare / sched_count, which (5) have no call and no scratch access (no stack frame, no spill
                             of its own) and does not touch the vector registers that hold the kernel's spilled scalars, so
                             nothing of the queue loop's state passes through memory or through a callee's epilogue; (6) the
                             queue's atomic is inside the lane-election s_and_saveexec / s_or_b64 exec pair(s) only; (7) the
                             second pass of a job (tail_pass, round 4) is the ONE call that is not a leaf: an ordinary function
                             with a frame of its own (as compress_one is for the other kernel: its prologue / epilogue save and
                             restore what it uses, EXEC is all ones at the call), which must not call anything itself and must
                             not touch the registers that hold the kernel's spilled scalars.
usage: python tools/check_queue_loops.py [extra hipcc flags]   (exit code 1 and a list of findings on failure)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_COMPRESS_KERNEL_INSTRUCTIONS = 600   # today ~250: queue loop + argument marshalling


def compile_isa(extra):
    out = os.path.join(tempfile.gettempdir(), "sq_api_queue_loops.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
           os.path.join(ROOT, "squishrs_amd", "csrc", "sq_api.hip")] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(2)
    return open(out).read().split("\n")


def body_of(lines, prefix):
    start = next(i for i, l in enumerate(lines) if re.match(r"^%s\S*:" % re.escape(prefix), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    name = lines[start].split(":")[0]
    return name, [l for l in lines[start + 1:end]]


def instrs(body):
    return [l.strip() for l in body if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]


def call_targets(body):
    """(line index, callee symbol) of every s_swappc_b64: the symbol of the `s_add_u32 sN, sN, SYM@rel32@lo+4` before it"""
    out = []
    for i, l in enumerate(body):
        if "s_swappc_b64" in l:
            sym = None
            for j in range(i - 1, max(0, i - 80), -1):
                m = re.search(r"s_add_u32\s+\S+\s+\S+\s+(\S+)@rel32@lo", body[j].replace(",", " "))
                if m:
                    sym = m.group(1)
                    break
            out.append((i, sym))
    return out


def atomic_region_ok(body, findings, kernel):
    """The queue's atomic add = the first global_atomic_add behind the kernel's outermost loop header.  It must sit in the
    lane-election region(s) only -- s_and_saveexec (lane 0 / first active lane), nothing else writing EXEC since the loop
    header -- and every region must be closed by `s_or_b64 exec, exec, ...` before the first branch that is not the
    regions' own s_cbranch_execz."""
    # (the queue fetch is the kernel's first RETURNING atomic add -- `sc0` --; its loop header is the nearest one above it)
    a = next((i for i, l in enumerate(body) if "global_atomic_add" in l and " sc0" in l), None)
    if a is None:
        findings.append(f"{kernel}: no returning global_atomic_add found (the queue loop changed shape: update this check)")
        return
    hdr = next((i for i in range(a, -1, -1) if "Loop Header" in body[i]), None)
    if hdr is None or a - hdr > 200:
        findings.append(f"{kernel}: no loop header within 200 lines above the queue's atomic add (update this check)")
        return
    opens = 0
    saved = False   # `s_mov_b64 sN, exec` seen: the expanded form of s_and_saveexec (mask taken from a spill lane)
    for j in range(hdr, a):
        t = body[j].strip()
        if t.startswith("s_and_saveexec_b64"):
            opens += 1
        elif re.match(r"s_mov_b64 s\[\d+:\d+\], exec", t):
            saved = True
        elif t.startswith("s_mov_b64 exec,") and saved:
            opens += 1
            saved = False
        elif re.match(r"s_\w+\s+exec", t) or ("saveexec" in t):
            findings.append(f"{kernel}: EXEC is written between the queue loop's header and its atomic add by something other than the lane election: {t}")
        elif t.startswith("s_cbranch") and not t.startswith("s_cbranch_execz"):
            # branches of the loop's own control (pass / list selection) are fine as long as no region is open
            if opens:
                findings.append(f"{kernel}: a branch inside the lane-election region in front of the atomic add: {t}")
    if not 1 <= opens <= 2:
        findings.append(f"{kernel}: the queue's atomic add is inside {opens} s_and_saveexec regions (expected the lane election: 1 or 2)")
    closes = 0
    for j in range(a + 1, min(len(body), a + 60)):
        t = body[j].strip()
        if t.startswith("s_or_b64 exec, exec"):
            closes += 1
            if closes == opens:
                break
        elif t.startswith("s_cbranch") and not t.startswith("s_cbranch_execz"):
            findings.append(f"{kernel}: a branch follows the queue's atomic add before EXEC is restored: {t}")
            break
        elif t.startswith("s_branch") or t.startswith("s_swappc"):
            findings.append(f"{kernel}: control leaves the election region before EXEC is restored: {t}")
            break
    if closes != opens:
        findings.append(f"{kernel}: EXEC is restored {closes} time(s) behind the queue's atomic add, {opens} region(s) were opened")


def main():
    lines = compile_isa(sys.argv[1:])
    findings = []
    # ---- k_compress_batch
    kname, kb = body_of(lines, "_ZN3sqz16k_compress_batch")
    calls = call_targets(kb)
    if len(calls) != 1 or not calls[0][1] or "compress_one" not in calls[0][1]:
        findings.append(f"k_compress_batch: expected exactly one call, to compress_one; found {[c[1] for c in calls]}")
    n = len(instrs(kb))
    if n > MAX_COMPRESS_KERNEL_INSTRUCTIONS:
        findings.append(f"k_compress_batch: {n} instructions in the kernel body (> {MAX_COMPRESS_KERNEL_INSTRUCTIONS}): the per-chunk body looks inlined")
    atomic_region_ok(kb, findings, "k_compress_batch")
    # ---- k_incompressible_batch
    _, fb = body_of(lines, "_ZN3sqz22k_incompressible_batch")
    fcalls = call_targets(fb)
    LEAVES = ("fast_sched_mask_rare", "sched_count")
    FRAMED = ("tail_pass",)
    bad = [c[1] for c in fcalls if not c[1] or not any(x in c[1] for x in LEAVES + FRAMED)]
    if bad:
        findings.append(f"k_incompressible_batch: calls other than the leaves {LEAVES} and {FRAMED} inside the persistent loop: {bad}")
    if sum(1 for c in fcalls if c[1] and "tail_pass" in c[1]) > 1:
        findings.append("k_incompressible_batch: more than one call site of tail_pass")
    atomic_region_ok(fb, findings, "k_incompressible_batch")
    wl = re.findall(r"v_writelane_b32 (v\d+),", "\n".join(fb))
    spill_vgprs = {v for v in set(wl) if wl.count(v) >= 8}   # the registers hipcc spills scalars into (a lone v_writelane is ordinary code)
    for leaf, prefix in (("fast_sched_mask_rare", "_ZN3sqz20fast_sched_mask_rare"), ("sched_count", "_ZN3sqz11sched_count")):
        try:
            _, lb = body_of(lines, prefix)
        except StopIteration:
            if any(c[1] and leaf in c[1] for c in fcalls):
                findings.append(f"k_incompressible_batch calls {leaf}, which this check cannot find")
            continue
        for t in instrs(lb):
            if t.startswith("s_swappc") or t.startswith("scratch_") or t.startswith("buffer_"):
                findings.append(f"{leaf}: not a leaf without a stack any more: " + t)
            for v in spill_vgprs:
                if re.search(r"\b%s\b" % v, t.split(";")[0]):
                    findings.append(f"{leaf}: touches {v}, which holds spilled scalars of k_incompressible_batch: " + t)
    try:
        _, tb = body_of(lines, "_ZN3sqz9tail_pass")
        for t in instrs(tb):
            if t.startswith("s_swappc"):
                findings.append("tail_pass: calls another function: " + t)
            for v in spill_vgprs:
                if re.search(r"\b%s\b" % v, t.split(";")[0]):
                    findings.append(f"tail_pass: touches {v}, which holds spilled scalars of k_incompressible_batch: " + t)
        if not any(t.startswith("s_setpc_b64 s[30:31]") for t in instrs(tb)):
            findings.append("tail_pass: no ordinary return found")
    except StopIteration:
        if any(c[1] and "tail_pass" in c[1] for c in fcalls):
            findings.append("k_incompressible_batch calls tail_pass, which this check cannot find")
    print(f"k_compress_batch: {n} instructions, calls {[c[1][:40] for c in calls if c[1]]}; "
          f"k_incompressible_batch: {len(fcalls)} call(s) to the leaf, spill registers {sorted(spill_vgprs)}; {len(findings)} finding(s)")
    for f in findings:
        print("FINDING:", f)
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
