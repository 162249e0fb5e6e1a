#!/usr/bin/env python3
"""Checks the gfx950 assembly for register-allocator traffic that hipcc placed under the WRONG lane mask.

The mechanism (found in round 4 by bisecting the faulting build of round 3 on its assembly, DESIGN.md section 5 and
profiles/r4_fault_bisect.txt): hipcc 7.2 (clang 22) may insert a live-range-split copy -- or a spill reload -- at
the head of a control-flow JOIN block, in front of the `s_or_b64 exec, exec, sX` that re-enables the lanes which
skipped the branch:

        s_and_saveexec_b64 s[2:3], s[60:61]        ; lanes 0..39 enter the block
        s_cbranch_execz .LBB9_368
        ...
    .LBB9_368:                                     ; %Flow
        v_mov_b32_e32 v78, v74                     ; <-- parks a value that is live in ALL lanes: copies lanes 0..39 only
        s_or_b64 exec, exec, s[2:3]
        ...                                        ; (v74 is reused)
        v_mov_b32_e32 v74, v78                     ; full mask: lanes 40..63 receive whatever v78 held before

The lanes that were masked off lose the value.  In the faulting build the value was a lane-derived index hoisted out
of the persistent item loop, the stale content was half of a 64-bit address, and the wavefront's SECOND item read
2.4 GB outside every buffer.  Nothing in the source is wrong; whether the allocator splits there depends on
register pressure (the two-wavefronts-per-SIMD builds) and on what is live across the region (anything loop
invariant that the compiler hoists over a whole solve).

Check 1 (exact, validated on the faulting build: flags that one instruction and nothing else): for every
lane-masking `s_and_saveexec` / `s_mov_b64 exec` / `s_xor_b64 exec` ... followed by `s_cbranch_execz LABEL`, no
VALU / VMEM / DS instruction may stand between LABEL and the instruction that restores EXEC (v_readlane /
v_writelane do not depend on EXEC and are allowed: that is SGPR spill traffic).

Check 2 (heuristic): a `scratch_load` reload executed at a lane-mask nesting depth d > 0 whose destination is read
at a depth < d before it is written again, or a spill slot stored at depth d and reloaded at a lower depth.  The
nesting depth is reconstructed from the text of structured code: +1 from `s_and_saveexec_b64 sX` (or `s_mov_b64 sA,
exec` ... `s_mov_b64 exec, sX`) to the matching `s_or_b64 exec, exec, sX`; +1 over the body of a lane-dropping loop
(`s_andn2_b64 exec, exec, sX` + `s_cbranch_execnz`); whole-wave sections (`s_or_saveexec_b64 sX, -1`) count as
unmasked.

usage: python scripts/check_spill_exec.py file.s [-v]       exit code 1 on a finding"""
import re
import sys

FUNC = re.compile(r"^(_Z\w+):")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def split_ops(ins):
    """(mnemonic, dest operand text, source operand text) -- dest = first operand for VALU / loads."""
    ins = ins.split(";")[0].strip()
    if not ins or ins.endswith(":") or ins.startswith("."):
        return None
    parts = ins.split(None, 1)
    mn = parts[0]
    ops = parts[1] if len(parts) > 1 else ""
    first, _, rest = ops.partition(",")
    return mn, first, rest


def depth_map(lines):
    """Lane-mask nesting depth of every line of one function (linear text order, structured code)."""
    depth = [0] * len(lines)
    stack = []                     # SGPR pairs that hold an outer mask
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    bump = [0] * (len(lines) + 1)  # extra depth over loop bodies (difference array)
    wwm = None
    last_saved = None              # `s_mov_b64 sA, exec`: the pair a later `s_mov_b64 exec, sX` narrows from
    for i, l in enumerate(lines):
        s = l.split(";")[0].strip()
        m = re.match(r"s_or_saveexec_b64 (s\[\d+:\d+\]), -1", s)
        if m:
            wwm = m.group(1)
        depth[i] = 0 if wwm else len(stack)
        if wwm and re.match(r"s_mov_b64 exec, " + re.escape(wwm), s):
            wwm = None
            continue
        m = re.match(r"s_and_saveexec_b64 (s\[\d+:\d+\]|vcc)", s)
        if m:
            stack.append(m.group(1))
            continue
        m = re.match(r"s_mov_b64 (s\[\d+:\d+\]), exec$", s)
        if m:
            last_saved = m.group(1)
            continue
        m = re.match(r"s_mov_b64 exec, (s\[\d+:\d+\])", s)
        if m:                                          # (the form hipcc uses when the condition mask was spilled)
            if m.group(1) in stack:
                del stack[stack.index(m.group(1)):]
            elif last_saved:
                stack.append(last_saved)
            continue
        m = re.match(r"s_or_saveexec_b64 (s\[\d+:\d+\]), (s\[\d+:\d+\])", s)
        if m:                                          # else: exec = saved | exec, the new pair keeps the then-mask
            if m.group(2) in stack:
                del stack[stack.index(m.group(2)):]
            stack.append(m.group(1))
            depth[i] = len(stack)
            continue
        m = re.match(r"s_or_b64 exec, exec, (s\[\d+:\d+\]|vcc)", s)
        if m:
            if m.group(1) in stack:
                del stack[stack.index(m.group(1)):]
            continue
        m = re.match(r"s_andn2_b64 exec, exec, (s\[\d+:\d+\])", s)
        if m and i + 1 < len(lines):
            b = re.match(r"\s*s_cbranch_execnz (\.LBB\d+_\d+)", lines[i + 1])
            if b and b.group(1) in labels and labels[b.group(1)] < i:
                bump[labels[b.group(1)]] += 1
                bump[i + 2] -= 1
    run = 0
    for i in range(len(lines)):
        run += bump[i]
        depth[i] += run
    return depth


NARROW = re.compile(r"s_(and|andn2|or|xor)_saveexec_b64|s_(mov|and|andn2|xor)_b64 exec\b")
EXECW = re.compile(r"s_\w+ exec\b|saveexec")
VECTOR = re.compile(r"(v_(?!readlane|writelane|readfirstlane)|scratch_|ds_|global_|buffer_|flat_)")


def join_findings(name, lines):
    """Check 1: vector instructions at the head of a join block, in front of the EXEC restore."""
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out, seen = [], set()
    for i, l in enumerate(lines):
        m = re.match(r"\s*s_cbranch_execz (\.LBB\d+_\d+)", l)
        if not m or m.group(1) not in labels or m.group(1) in seen:
            continue
        k = i - 1
        while k >= 0 and not lines[k].split(";")[0].strip():
            k -= 1
        if not NARROW.match(lines[k].split(";")[0].strip()):
            continue                                    # (hipcc also writes `s_cbranch_execz` beside uniform branches)
        seen.add(m.group(1))
        tgt, bad, restored = labels[m.group(1)], [], False
        for j in range(tgt + 1, min(len(lines), tgt + 60)):
            t = lines[j].split(";")[0].strip()
            if re.match(r"^\.LBB", lines[j]):
                break
            if not t or t.startswith("."):
                continue
            if EXECW.match(t):
                restored = True
                break
            if re.match(r"s_(cbranch|branch|endpgm|setpc|swappc|barrier)", t):
                break
            if VECTOR.match(t):
                bad.append((j, t))
        if restored:
            for j, t in bad:
                out.append((name, j, "join %s: `%s` (line %d) stands in front of the EXEC restore" % (m.group(1), t[:60], j + 1)))
    return out


def check_function(name, lines, verbose):
    depth = depth_map(lines)
    findings = join_findings(name, lines)
    nload = nstore = 0
    slots = {}                                          # offset -> max depth it was stored at
    for i, l in enumerate(lines):
        op = split_ops(l)
        if not op:
            continue
        mn, first, rest = op
        if mn.startswith("scratch_store"):
            nstore += 1
            off = re.search(r"offset:(\d+)", l)
            key = int(off.group(1)) if off else 0
            slots.setdefault(key, []).append((i, depth[i]))
    for i, l in enumerate(lines):
        op = split_ops(l)
        if not op:
            continue
        mn, first, rest = op
        if not mn.startswith("scratch_load"):
            continue
        nload += 1
        d = depth[i]
        off = re.search(r"offset:(\d+)", l)
        key = int(off.group(1)) if off else 0
        for (si, sd) in slots.get(key, []):
            if sd > d:
                findings.append((name, si, "slot %d stored at mask depth %d (line %d) and reloaded at depth %d (line %d)"
                                 % (key, sd, si + 1, d, i + 1)))
        if d == 0:
            continue
        live = vregs(first)
        for j in range(i + 1, len(lines)):
            o2 = split_ops(lines[j])
            if not o2 or not live:
                if not live:
                    break
                continue
            mn2, f2, r2 = o2
            reads = vregs(r2)
            writes = vregs(f2)
            if mn2.startswith(("scratch_store", "global_store", "flat_store", "ds_write", "buffer_store", "v_cmp", "v_writelane")):
                reads |= writes
                writes = set()
            elif re.match(r"v_(fmac|mac)|v_cndmask.*_dpp|v_mov_b\d+_dpp|v_readlane|v_readfirstlane", mn2):
                reads |= writes                         # read-modify-write / partial writers
                if mn2.startswith(("v_readlane", "v_readfirstlane")):
                    writes = set()
            hit = reads & live
            if hit and depth[j] < d:
                findings.append((name, i, "reload at mask depth %d (line %d: %s) read at depth %d (line %d: %s)"
                                 % (d, i + 1, l.strip()[:60], depth[j], j + 1, lines[j].strip()[:60])))
                live -= hit
            if depth[j] <= d and not re.match(r"v_(fmac|mac)|v_\w+_dpp|v_writelane", mn2):
                live -= writes                          # overwritten under a mask at least as wide
            if re.match(r"s_endpgm|s_setpc_b64", mn2):
                break
    if verbose or findings:
        print("%s: %d scratch loads, %d scratch stores, %d finding(s)" % (name, nload, nstore, len(findings)))
    for f in findings[:40]:
        print("   ", f[2])
    return nload, nstore, len(findings)


def main():
    path = sys.argv[1]
    verbose = "-v" in sys.argv
    text = open(path).read().splitlines()
    funcs, cur, start = [], None, 0
    for n, line in enumerate(text):
        m = FUNC.match(line)
        if m:
            if cur:
                funcs.append((cur, text[start:n]))
            cur, start = m.group(1), n
    if cur:
        funcs.append((cur, text[start:]))
    tot = [0, 0, 0]
    for name, lines in funcs:
        end = next((k for k, l in enumerate(lines) if re.match(r"\s*(s_endpgm|s_setpc_b64)", l)), len(lines) - 1)
        r = check_function(name, lines[:end + 1], verbose)
        tot = [a + b for a, b in zip(tot, r)]
    print("%d functions checked: %d scratch reloads, %d scratch stores, %d finding(s)" % (len(funcs), tot[0], tot[1], tot[2]))
    return 1 if tot[2] else 0


if __name__ == "__main__":
    sys.exit(main())
